"""The 80 COCO detection class names, ids 1..80 (label table `data_meta` hands to the visualiser; reference:
crowdsam/coco_names.py).  Public dataset vocabulary; only the 'person' entry matters on the Crowd-SAM path."""
_NAMES = """person bicycle car motorcycle airplane bus train truck boat traffic_light fire_hydrant stop_sign
parking_meter bench bird cat dog horse sheep cow elephant bear zebra giraffe backpack umbrella handbag tie suitcase
frisbee skis snowboard sports_ball kite baseball_bat baseball_glove skateboard surfboard tennis_racket bottle
wine_glass cup fork knife spoon bowl banana apple sandwich orange broccoli carrot hot_dog pizza donut cake chair
couch potted_plant bed dining_table toilet tv laptop mouse remote keyboard cell_phone microwave oven toaster sink
refrigerator book clock vase scissors teddy_bear hair_drier toothbrush""".split()
coco_classes = {i + 1: n.replace("_", " ").capitalize() for i, n in enumerate(_NAMES)}
assert len(coco_classes) == 80
