"""ref src/models/components/struct_graph_encoder.py surface -> HIP implementation."""
from oneprot_amd.encoders import StructEncoder  # noqa: F401
