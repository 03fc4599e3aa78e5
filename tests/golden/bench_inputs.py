"""Constants shared by the generator of the benchmark-configuration fixtures (make_golden.py: c2_batch32.npz,
c4_pairs.npz, c2_fp64_order.npz -- build container, runs the reference) and the tests / tools that read them on the GPU
box.  No reference code involved."""
import torch

DESC_SAMPLE_ROWS = 32  # descriptor rows stored in full per image; every row also gets two weighted checksums
C4_PAIRS = (0, 12, 21, 31)  # pairs of synthetic_pairs(32, 1024, 1024, seed=1234) held by c4_pairs.npz
FP64_IMAGES = 16  # views 0 and 1 of pairs 0..7 of the C2 batch
FP64_DEPTH = 1152  # ranks kept of the exact order (1024 selected + the first 128 beyond the boundary)


def desc_checksum_weights(dim=256):
    """Two fixed weight vectors (|w| <= 1) for the per-row descriptor checksums."""
    c = torch.arange(dim, dtype=torch.float64)
    return torch.stack([torch.cos(0.37 * c + 0.11), torch.sin(0.23 * c * c + 0.5)]).float()
