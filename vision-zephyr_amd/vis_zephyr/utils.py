"""ref:vis_zephyr/utils.py:6-10."""


def disable_torch_init():
    """The reference skips torch.nn parameter initialisation to load faster.  The MI355X engine never
    builds torch.nn modules, so there is nothing to disable; kept because callers invoke it
    (ref:vis_zephyr/serve/cli.py:51, ref:vis_zephyr/eval/eval_vqa.py:135)."""
    return None
