import torch
print("priority range (least, greatest):", torch.cuda.Stream.priority_range())
for p in (-2,-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print(p, "->", s.priority)
    except Exception as e: print(p, "err", e)
