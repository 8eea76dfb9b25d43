"""Host-side mirror of the reference's `models` package: `from models.sit import SiT`,
`from models.mpp import masked_patch_pretraining` resolve here when this package directory is put
on sys.path ahead of the reference's root."""
