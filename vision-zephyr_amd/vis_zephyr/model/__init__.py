import os as _os

from .language_model.vis_zephyr import VisZephyrForCausalLM, VisZephyrConfig  # noqa: F401

_ref = _os.environ.get("VIS_ZEPHYR_REFERENCE")
if _ref and _os.path.isdir(_os.path.join(_ref, "vis_zephyr", "model")):
    __path__.append(_os.path.join(_ref, "vis_zephyr", "model"))     # vip_processor (host-side drawing) stays the reference's
