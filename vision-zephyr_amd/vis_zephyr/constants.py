"""Names and sentinel values callers import from `vis_zephyr.constants` (values as in ref:vis_zephyr/constants.py:5-20).

Two of them are load-bearing on the hot path: IMAGE_TOKEN_INDEX marks, inside `input_ids`, where the 32 x N visual tokens of a
request are spliced in (`prepare_inputs_labels_for_multimodal`, the engine's `vz_embed_splice`), and IGNORE_INDEX is the label
value the loss skips (every spliced visual position gets it).  The token strings are what `tokenizer_image_token` and the
conversation templates look for; the remaining three are read by the reference's serving scripts only."""

# ---- sentinels inside id / label tensors ----
IMAGE_TOKEN_INDEX = -200
IGNORE_INDEX = -100

# ---- prompt-side markers ----
DEFAULT_IMAGE_TOKEN = "<image>"
IMAGE_PLACEHOLDER = "<image-placeholder>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"

# ---- serving scripts (not used by this package) ----
LOGDIR = "."
WORKER_HEART_BEAT_INTERVAL = 15
CONTROLLER_HEART_BEAT_EXPIRATION = 30
