"""`_target_: src.models.components.sequence_encoder.SequenceEncoder` (ref configs/model/components/sequence.yaml:2)."""
from oneprot_amd.encoders import SequenceEncoder  # noqa: F401
