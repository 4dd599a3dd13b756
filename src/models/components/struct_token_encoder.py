"""`_target_: src.models.components.struct_token_encoder.StructTokenEncoder` (ref configs/model/components/struct_token.yaml:2)."""
from oneprot_amd.encoders import StructTokenEncoder  # noqa: F401
