"""Drop-in mirror of the reference's ``segment_anything_cs`` package surface backed
by the MI355X-native HIP kernels in ``crowdsam_amd``.  Same names, arguments and error behaviour for
the Crowd-SAM dense-prompt path; everything the reference leaves broken or unused is not mirrored."""
from .build_sam import (build_sam, build_sam_vit_b, build_sam_vit_h, build_sam_vit_l,  # noqa: F401
                        sam_model_registry)
from .predictor import SamPredictor  # noqa: F401
