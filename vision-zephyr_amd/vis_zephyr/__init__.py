"""Drop-in `vis_zephyr` package for the MI355X build: same import paths as the reference's package
(ref:vis_zephyr/__init__.py, ref:vis_zephyr/model/__init__.py:1-4), arithmetic in libviszephyr_hip.so."""
