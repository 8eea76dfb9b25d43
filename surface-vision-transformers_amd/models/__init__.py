"""Host-side mirror of the reference's `models` package.

Normal use: `import sitk; from sitk.models.sit import SiT`.

Drop-in use: with this package's parent directory ahead of the reference's root on
sys.path, the reference tools' `from models.sit import SiT` / `from models.mpp import
masked_patch_pretraining` resolve here.  In that case this package was imported under the top-level
name `models`; alias its submodules to the canonical `sitk.models.*` modules so that there is exactly
one `SiT` class (engine.py type checks, pickling) whichever name was used.
"""
import importlib
import sys

if __name__ == "models":
    import sitk  # noqa: F401  (repo root must be on sys.path as well)

    for _sub in ("sit", "mpp"):
        sys.modules[f"models.{_sub}"] = importlib.import_module(f"sitk.models.{_sub}")
