from .openpose import OpenPose, batches, read_img_u8
from .preprocess import crop_and_normalize
from .structures import Target
