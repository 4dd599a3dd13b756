"""`_target_: src.models.oneprot_module.OneProtLitModule` (ref configs/model/default.yaml:1) -> HIP implementation."""
from oneprot_amd.module import OneProtLitModule  # noqa: F401
