"""ref src/models/components/base_encoder.py surface -> HIP implementation."""
from oneprot_amd.encoders import (Attention1dPooling, BaseEncoder, CLSTokenPooling, LearnableLogitScaling, MaskedConv1d, MeanPooling,  # noqa: F401
                                  Normalize)
