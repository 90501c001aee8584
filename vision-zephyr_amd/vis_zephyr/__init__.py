"""Drop-in `vis_zephyr` package for the MI355X build: same import paths as the reference's package
(ref:vis_zephyr/__init__.py, ref:vis_zephyr/model/__init__.py:1-4), arithmetic in libviszephyr_hip.so.

The callers that sit above the boundary (`vis_zephyr.serve.cli`, `vis_zephyr.eval.*`, `vis_zephyr.model.vip_processor`)
are not re-implemented here.  To run them unchanged (`script/run_cli.sh`, `script/eval/eval_qa.sh`) point
VIS_ZEPHYR_REFERENCE at a checkout of the reference: its package directory is appended to this package's search
path, so every module that exists here (model, constants, conversation, utils, mm_utils, ...) resolves to the MI355X
implementation and only the caller scripts resolve to the reference's files."""
import os as _os

_ref = _os.environ.get("VIS_ZEPHYR_REFERENCE")
if _ref and _os.path.isdir(_os.path.join(_ref, "vis_zephyr")):
    __path__.append(_os.path.join(_ref, "vis_zephyr"))
