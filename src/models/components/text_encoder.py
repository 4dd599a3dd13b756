"""`_target_: src.models.components.text_encoder.TextEncoder` (ref configs/model/components/text.yaml:2)."""
from oneprot_amd.encoders import TextEncoder  # noqa: F401
