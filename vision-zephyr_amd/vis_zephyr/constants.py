"""Constants that are part of the model API (ref:vis_zephyr/constants.py:5-20)."""
CONTROLLER_HEART_BEAT_EXPIRATION = 30
WORKER_HEART_BEAT_INTERVAL = 15
LOGDIR = "."

IGNORE_INDEX = -100        # label value excluded from the loss
IMAGE_TOKEN_INDEX = -200   # sentinel id marking where an image's visual tokens are spliced in

DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"
IMAGE_PLACEHOLDER = "<image-placeholder>"
