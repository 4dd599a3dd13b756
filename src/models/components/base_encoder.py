"""ref src/models/components/base_encoder.py surface -> HIP implementation."""
from oneprot_amd.encoders import BaseEncoder, CLSTokenPooling, LearnableLogitScaling, MeanPooling, Normalize  # noqa: F401
