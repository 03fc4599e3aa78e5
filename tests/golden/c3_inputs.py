"""Inputs of the BASELINE config-3 fixture (tests/golden/pipeline_official.npz): HPatches-shaped synthetic RGB pairs,
regenerated bit for bit from seeds by both the generator of the golden vectors (make_golden.py, build container, runs
the reference) and the GPU test (tests/test_gpu_c3.py).  No reference code involved."""
import torch

from glue_factory_colon_amd import synthetic

C3_PAIRS = [  # (name, seed, (h0, w0), (h1, w1), original (w, h) of both views): short side 480, arbitrary long side
    ("v_synth0/2.ppm", 301, (480, 613), (480, 640), ((800, 626), (1000, 750))),
    ("v_synth1/3.ppm", 302, (480, 640), (725, 480), ((1280, 960), (640, 967))),
    ("i_synth2/5.ppm", 303, (725, 480), (480, 613), ((768, 1160), (1024, 802))),
]

# Round 4: five more pairs under `detection_threshold` 0.5 (fixture pipeline_official_ragged.npz).  With the name-seeded
# weights a VGA-class view has ~12 000 NMS maxima above 0 and its 1024th score is 0.47..0.52, so at 0.5 the views keep
# between ~850 and 1024 (the cap) key points: the official extractor's ragged path (gluefactory_nonfree/superpoint.py:
# 267-300,349-370 -- fewer than k detections, different counts in the two views) at config-3 sizes.
C3_RAGGED_THRESHOLD = 0.5
C3_RAGGED_PAIRS = [
    ("v_synth3/2.ppm", 304, (480, 640), (480, 640), ((1024, 768), (1024, 768))),
    ("v_synth3/4.ppm", 305, (480, 640), (480, 613), ((1024, 768), (800, 626))),
    ("i_synth4/3.ppm", 306, (480, 656), (480, 656), ((984, 720), (984, 720))),
    ("i_synth4/6.ppm", 307, (640, 480), (725, 480), ((768, 1024), (640, 967))),
    ("v_synth5/5.ppm", 308, (480, 613), (480, 613), ((800, 626), (800, 626))),
]


def c3_pair(seed, size0, size1, origs):
    """HPatches-shaped synthetic RGB pair (uint8-quantised, exactly regenerable from the seed): two crops of one
    band-limited canvas, displaced by (24, 16) pixels, so that true correspondences exist.  Returns the data dict of
    one loader item as datasets/hpatches.py + utils/image.py:15-72 produce it (image [1,3,H,W] in [0,1], `scales` =
    new / original size, `image_size` (w, h))."""
    canvas = synthetic.synthetic_images(1, 760, 680, seed=seed)[0, 0]

    def view(h, w, y, x, orig):
        g = canvas[y:y + h, x:x + w]
        rgb = torch.stack([g * 0.8, g, g * 0.9], 0).clamp(0, 1)
        rgb = ((rgb * 255).round() / 255).float()[None]  # what a decoded uint8 image gives
        scales = torch.tensor([[w / orig[0], h / orig[1]]], dtype=torch.float32)
        return {"image": rgb, "image_size": torch.tensor([[float(w), float(h)]]), "scales": scales,
                "original_image_size": torch.tensor([[float(orig[0]), float(orig[1])]])}

    return {"view0": view(*size0, 0, 0, origs[0]), "view1": view(*size1, 16, 24, origs[1])}
