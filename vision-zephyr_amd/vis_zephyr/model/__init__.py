from .language_model.vis_zephyr import VisZephyrForCausalLM, VisZephyrConfig  # noqa: F401
