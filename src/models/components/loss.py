"""ref src/models/components/loss.py surface -> HIP / RCCL implementation."""
from oneprot_amd.loss import (ClipLoss, NeighbourExchange, NeighbourExchangeBidir, SigLipLoss, gather_features, neighbour_exchange,  # noqa: F401
                              neighbour_exchange_bidir, neighbour_exchange_bidir_with_grad, neighbour_exchange_with_grad)
