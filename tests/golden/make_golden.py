"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container, where /root/reference exists (it does not travel to
the GPU box; the tests read the committed .npz files, never the reference).  The
reference needs `omegaconf`, which is absent here: `tests/golden/_standins/omegaconf`
is a ~100-line stand-in written for this script.  No network access is
made: weights come from glue_factory_colon_amd.weights (name-seeded), the open
SuperPoint takes them through its local-file branch (superpoint_open.py:120-121), the
in-tree LightGlue through load_state_dict, and for the official SuperPoint class the
URL fetch in its constructor is redirected to the same generated state dict.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures hold inputs and reference outputs only (data), in float32 / int64.
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("GFC_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_standins"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from glue_factory_colon_amd import synthetic, weights  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KB, keys={sorted(arrays)}")


# ------------------------------------------------------------------ reference modules
from gluefactory.models.extractors import superpoint_open as ref_spo  # noqa: E402
from gluefactory.models.matchers import lightglue as ref_lg  # noqa: E402
from gluefactory.models.two_view_pipeline import TwoViewPipeline  # noqa: E402
import gluefactory_nonfree.superpoint as ref_sp  # noqa: E402

_tmp = tempfile.mkdtemp()
SPO_PATH = os.path.join(_tmp, "spo_seed0.pth")
torch.save(weights.superpoint_open_state_dict(0), SPO_PATH)


def make_spo(**conf):
    return ref_spo.SuperPoint({"weights": SPO_PATH, **conf}).eval()


def make_sp_official(**conf):
    sd = weights.superpoint_state_dict(0)
    orig = torch.hub.load_state_dict_from_url
    torch.hub.load_state_dict_from_url = lambda *a, **k: sd  # no network: constructor fetch redirected
    try:
        m = ref_sp.SuperPoint(conf).eval()
    finally:
        torch.hub.load_state_dict_from_url = orig
    return m


def make_lg(input_dim=256, **conf):
    m = ref_lg.LightGlue({"input_dim": input_dim, **conf}).eval()
    missing = m.load_state_dict(weights.lightglue_state_dict(0, input_dim=input_dim), strict=False)
    assert missing.missing_keys == ["confidence_thresholds"] and not missing.unexpected_keys, missing
    return m


# --------------------------------------------------------------------------- NMS cases
def golden_nms():
    g = torch.Generator().manual_seed(7)
    out = {}
    for r in (0, 1, 3, 4):
        s = torch.rand((2, 45, 70), generator=g)
        # plateaus and exact ties: quantise one image, add a constant patch and zeros
        s[1] = (s[1] * 8).round() / 8
        s[0, 10:14, 20:30] = 0.75
        s[0, 30:, :5] = 0.0
        out[f"in_r{r}"] = npy(s)
        out[f"out_r{r}"] = npy(ref_spo.batched_nms(s, r))
        out[f"out_official_r{r}"] = npy(ref_sp.simple_nms(s, r))
    save("nms", **out)


# ----------------------------------------------------------------------- SuperPoint-open
def golden_superpoint_open():
    img = synthetic.synthetic_images(2, 120, 160, seed=11)
    out = {"image": npy(img)}
    # stage tensors through the reference's own sub-modules
    m = make_spo(max_num_keypoints=150, detection_threshold=0.0, nms_radius=3, dense_outputs=True)
    feats = m.backbone(img)
    logits = m.detector(feats)
    out["logits"] = npy(logits)
    heat = torch.softmax(logits, 1)[:, :-1]
    b, _, h, w = heat.shape
    heat = heat.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
    out["heatmap"] = npy(heat)
    out["nms_r3"] = npy(ref_spo.batched_nms(heat, 3))
    for i in range(2):  # b == 1 calls (the eval configuration)
        p = m({"image": img[i:i + 1]})
        out[f"k150_kpts_{i}"] = npy(p["keypoints"][0])
        out[f"k150_scores_{i}"] = npy(p["keypoint_scores"][0])
        out[f"k150_desc_{i}"] = npy(p["descriptors"][0])
        if i == 0:
            out["dense_desc_0"] = npy(p["dense_descriptors"][0])
    # fewer detections than k: all candidates, row-major, unsorted (superpoint_open.py:54-58)
    m2 = make_spo(max_num_keypoints=4096, detection_threshold=0.0, nms_radius=4)
    p = m2({"image": img[:1]})
    out["k4096_r4_kpts_0"] = npy(p["keypoints"][0])
    out["k4096_r4_scores_0"] = npy(p["keypoint_scores"][0])
    out["k4096_r4_desc_0"] = npy(p["descriptors"][0])
    # threshold > 0, no NMS radius 0 (scripts/eval_hpatches.sh uses nms_radius 0), RGB input
    rgb = torch.cat([img[:1] * 0.9, img[:1], img[:1] * 0.8], 1).clamp(0, 1)
    out["image_rgb"] = npy(rgb)
    m3 = make_spo(max_num_keypoints=100, detection_threshold=0.02, nms_radius=0, remove_borders=6)
    p = m3({"image": rgb})
    out["rgb_r0_kpts"] = npy(p["keypoints"][0])
    out["rgb_r0_scores"] = npy(p["keypoint_scores"][0])
    out["rgb_r0_desc"] = npy(p["descriptors"][0])
    # batched call with force_num_keypoints (every image has >= k detections: no random padding)
    m4 = make_spo(max_num_keypoints=64, detection_threshold=0.0, nms_radius=3, force_num_keypoints=True)
    p = m4({"image": img})
    out["b2_k64_kpts"] = npy(p["keypoints"])
    out["b2_k64_scores"] = npy(p["keypoint_scores"])
    out["b2_k64_desc"] = npy(p["descriptors"])
    save("superpoint_open", **out)


def _specular_masks(b, h, w, seed):
    """Blobby boolean masks (True = keep), like specular-highlight masks: ~25 % of the pixels dropped."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand((b, 1, h // 8 + 1, w // 8 + 1), generator=g)
    m = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=False) > 0.42
    return m


def golden_specular():
    """The Endomapper addition of this reference: key points filtered by data["specular_mask"]
    (extractors/utils.py:4-42), before top-k in superpoint_open.py:177-188, after top-k in superpoint.py:310-328."""
    img = synthetic.synthetic_images(2, 120, 160, seed=21)
    masks = _specular_masks(2, 120, 160, 5)
    out = {"image": npy(img), "mask": npy(masks)}
    m = make_spo(max_num_keypoints=150, detection_threshold=0.0, nms_radius=3)
    for i in range(2):
        p = m({"image": img[i:i + 1], "specular_mask": masks[i:i + 1]})
        out[f"open_k150_kpts_{i}"] = npy(p["keypoints"][0])
        out[f"open_k150_scores_{i}"] = npy(p["keypoint_scores"][0])
        out[f"open_k150_desc_{i}"] = npy(p["descriptors"][0])
    # image_size crops the mask (w, h): key points beyond it are dropped; mask as [B,H,W] floats
    size = torch.tensor([[131.0, 97.0]])
    p = m({"image": img[:1], "specular_mask": masks[:1, 0].float(), "image_size": size})
    out["open_crop_size"] = npy(size)
    out["open_crop_kpts"] = npy(p["keypoints"][0])
    out["open_crop_scores"] = npy(p["keypoint_scores"][0])
    # batched + force_num_keypoints with fewer survivors than k is random padding (not a vector); k small enough here
    m4 = make_spo(max_num_keypoints=48, detection_threshold=0.0, nms_radius=3, force_num_keypoints=True)
    p = m4({"image": img, "specular_mask": masks})
    out["open_b2_k48_kpts"] = npy(p["keypoints"])
    out["open_b2_k48_scores"] = npy(p["keypoint_scores"])
    # official arithmetic: filter AFTER top-k (fewer than k key points come back)
    mo = make_sp_official(max_num_keypoints=150, detection_threshold=0.0005, nms_radius=3)
    for i in range(2):
        p = mo({"image": img[i:i + 1], "specular_mask": masks[i:i + 1]})
        out[f"off_k150_kpts_{i}"] = npy(p["keypoints"][0])
        out[f"off_k150_scores_{i}"] = npy(p["keypoint_scores"][0])
        out[f"off_k150_desc_{i}"] = npy(p["descriptors"][0])
    # soft-argmax refinement (superpoint.py:100-116,302-305): fractional key points, then the 4-pixel mask test
    mr = make_sp_official(max_num_keypoints=150, detection_threshold=0.0005, nms_radius=3, refinement_radius=2)
    p = mr({"image": img[:1]})
    out["refine_kpts"] = npy(p["keypoints"][0])
    out["refine_scores"] = npy(p["keypoint_scores"][0])
    out["refine_desc"] = npy(p["descriptors"][0])
    p = mr({"image": img[1:2], "specular_mask": masks[1:2]})
    out["refine_spec_kpts"] = npy(p["keypoints"][0])
    out["refine_spec_scores"] = npy(p["keypoint_scores"][0])
    save("specular", **out)


# ------------------------------------------------------------------- SuperPoint official
def golden_superpoint_official():
    img = synthetic.synthetic_images(1, 104, 136, seed=21)
    out = {"image": npy(img)}
    m = make_sp_official(max_num_keypoints=120, detection_threshold=0.0, nms_radius=3, sparse_outputs=False)
    p = m({"image": img})
    out["heatmap"] = npy(p["keypoint_scores"])
    out["dense_desc"] = npy(p["descriptors"])
    for legacy in (True, False):
        m = make_sp_official(max_num_keypoints=120, detection_threshold=0.0, nms_radius=3, legacy_sampling=legacy)
        p = m({"image": img})
        tag = "legacy" if legacy else "fixed"
        out[f"{tag}_kpts"] = npy(p["keypoints"][0])
        out[f"{tag}_scores"] = npy(p["keypoint_scores"][0])
        out[f"{tag}_desc"] = npy(p["descriptors"][0])
    # image_size smaller than the tensor: right / bottom borders follow the true extent
    size = torch.tensor([[120.0, 90.0]])
    m = make_sp_official(max_num_keypoints=-1, detection_threshold=0.01, nms_radius=4)
    p = m({"image": img, "image_size": size})
    out["sized_image_size"] = npy(size)
    out["sized_kpts"] = npy(p["keypoints"][0])
    out["sized_scores"] = npy(p["keypoint_scores"][0])
    out["sized_desc"] = npy(p["descriptors"][0])
    save("superpoint_official", **out)


# ---------------------------------------------------------------------------- LightGlue
def _features(n_img, h, w, k, seed):
    """Run reference SuperPoint-open on synthetic pairs to obtain realistic LightGlue inputs."""
    v0, v1 = synthetic.synthetic_pairs(n_img, h, w, seed=seed, dx=16, dy=8)
    m = make_spo(max_num_keypoints=k, detection_threshold=0.0, nms_radius=3, force_num_keypoints=True)
    return v0, v1, m({"image": v0}), m({"image": v1})


def golden_lightglue():
    h, w, k = 120, 160, 160
    v0, v1, p0, p1 = _features(2, h, w, k, seed=31)
    size = torch.tensor([[w, h]] * 2, dtype=torch.float32)
    data = {"keypoints0": p0["keypoints"], "keypoints1": p1["keypoints"],
            "descriptors0": p0["descriptors"], "descriptors1": p1["descriptors"],
            "view0": {"image_size": size}, "view1": {"image_size": size}}
    out = {k_: npy(v) for k_, v in data.items() if not k_.startswith("view")}
    out["image_size"] = npy(size)
    lg = make_lg(filter_threshold=0.1)
    # per-layer descriptors through the reference's own layer modules
    k0 = ref_lg.normalize_keypoints(data["keypoints0"], size).clone()
    k1 = ref_lg.normalize_keypoints(data["keypoints1"], size).clone()
    e0, e1 = lg.posenc(k0), lg.posenc(k1)
    out["enc0"] = npy(e0)
    d0, d1 = data["descriptors0"], data["descriptors1"]
    s0 = lg.transformers[0].self_attn(d0, e0)
    out["layer0_self0"] = npy(s0)
    for i in range(9):
        d0, d1 = lg.transformers[i](d0, d1, e0, e1)
        if i in (0, 4):
            out[f"layer{i}_desc0"] = npy(d0)
            out[f"layer{i}_desc1"] = npy(d1)
    pred = lg(data)
    for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "log_assignment",
                "ref_descriptors0", "ref_descriptors1", "prune0"):
        out["b2_" + key] = npy(pred[key])
    # threshold 0.0 (the module default)
    pred = make_lg(filter_threshold=0.0)(data)
    out["th0_matches0"] = npy(pred["matches0"])
    out["th0_matches1"] = npy(pred["matches1"])
    # ragged pair, b == 1: M != N
    data1 = {"keypoints0": p0["keypoints"][:1, :100], "keypoints1": p1["keypoints"][:1],
             "descriptors0": p0["descriptors"][:1, :100], "descriptors1": p1["descriptors"][:1],
             "view0": {"image_size": size[:1]}, "view1": {"image_size": size[:1]}}
    pred = lg(data1)
    for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "log_assignment"):
        out["ragged_" + key] = npy(pred[key])
    # 128-d descriptors (DISK-style input, input_proj Linear 128->256, lightglue.py:352-355)
    g = torch.Generator().manual_seed(5)
    dd0 = torch.nn.functional.normalize(torch.randn((1, 96, 128), generator=g), dim=-1)
    perm = torch.randperm(96, generator=g)
    dd1 = torch.nn.functional.normalize(dd0[:, perm] + 0.05 * torch.randn((1, 96, 128), generator=g), dim=-1)
    kk0 = torch.rand((1, 96, 2), generator=g) * torch.tensor([w, h])
    kk1 = kk0[:, perm] + torch.tensor([3.0, -2.0])
    lg128 = make_lg(input_dim=128, filter_threshold=0.1)
    pred = lg128({"keypoints0": kk0, "keypoints1": kk1, "descriptors0": dd0, "descriptors1": dd1,
                  "view0": {"image_size": size[:1]}, "view1": {"image_size": size[:1]}})
    out.update(d128_keypoints0=npy(kk0), d128_keypoints1=npy(kk1), d128_descriptors0=npy(dd0),
               d128_descriptors1=npy(dd1))
    for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "log_assignment"):
        out["d128_" + key] = npy(pred[key])
    save("lightglue", **out)


def golden_assignment():
    """sigmoid_log_double_softmax + filter_matches on crafted inputs with exact ties, a
    threshold edge and the empty-set early return (lightglue.py:257-269,294-319)."""
    g = torch.Generator().manual_seed(3)
    sim = torch.randn((2, 37, 53), generator=g) * 3
    sim[0, 5] = sim[0, 4]  # duplicate rows -> tied column maxima
    sim[1, :, 7] = sim[1, :, 6]  # duplicate columns -> tied row maxima
    sim = (sim * 4).round() / 4
    z0 = torch.randn((2, 37, 1), generator=g)
    z1 = torch.randn((2, 53, 1), generator=g)
    la = ref_lg.sigmoid_log_double_softmax(sim, z0, z1)
    out = {"sim": npy(sim), "z0": npy(z0), "z1": npy(z1), "log_assignment": npy(la)}
    for th in (0.0, 0.1, 0.5):
        m0, m1, s0, s1 = ref_lg.filter_matches(la, th)
        tag = str(th).replace(".", "p")
        out.update({f"m0_{tag}": npy(m0), f"m1_{tag}": npy(m1), f"s0_{tag}": npy(s0), f"s1_{tag}": npy(s1)})
    # a permutation-like assignment with large margins: everything matches
    perm = torch.randperm(40, generator=g)
    sim2 = torch.full((1, 40, 40), -5.0)
    sim2[0, torch.arange(40), perm] = 12.0
    la2 = ref_lg.sigmoid_log_double_softmax(sim2, torch.full((1, 40, 1), 4.0), torch.full((1, 40, 1), 4.0))
    m0, m1, s0, s1 = ref_lg.filter_matches(la2, 0.1)
    out.update(perm_log_assignment=npy(la2), perm_m0=npy(m0), perm_m1=npy(m1), perm_s0=npy(s0), perm_s1=npy(s1))
    m0, m1, s0, s1 = ref_lg.filter_matches(torch.zeros((2, 1, 6)), 0.1)
    out.update(empty_m0=npy(m0), empty_m1=npy(m1), empty_s0=npy(s0), empty_s1=npy(s1))
    save("assignment", **out)


# ---------------------------------------------------------------------------- pipeline
def golden_pipeline():
    """TwoViewPipeline (two_view_pipeline.py:278-405) on a synthetic pair and on the
    boat pair of tests/test_integration.py (down-sampled grey crop stored as uint8)."""
    out = {}
    conf = {"extractor": {"name": "extractors.superpoint_open", "weights": SPO_PATH, "max_num_keypoints": 256,
                          "detection_threshold": 0.0, "nms_radius": 3},
            "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1, "flash": False}}
    pipe = TwoViewPipeline(conf).eval()
    pipe.matcher.load_state_dict(weights.lightglue_state_dict(0), strict=False)
    v0, v1 = synthetic.synthetic_pairs(1, 160, 208, seed=41, dx=16, dy=8)
    size = torch.tensor([[208.0, 160.0]])
    pred = pipe({"view0": {"image": v0, "image_size": size}, "view1": {"image": v1, "image_size": size}})
    out["syn_image0"] = (npy(v0) * 255).round().astype(np.uint8)  # inputs stored as uint8: exact in fp32 /255
    out["syn_image1"] = (npy(v1) * 255).round().astype(np.uint8)
    v0q = torch.from_numpy(out["syn_image0"].astype(np.float32) / 255)
    v1q = torch.from_numpy(out["syn_image1"].astype(np.float32) / 255)
    pred = pipe({"view0": {"image": v0q, "image_size": size}, "view1": {"image": v1q, "image_size": size}})
    keys = ("keypoints0", "keypoints1", "keypoint_scores0", "keypoint_scores1", "descriptors0", "descriptors1",
            "matches0", "matches1", "matching_scores0", "matching_scores1")
    for key in keys:
        out["syn_" + key] = npy(pred[key])
    out["syn_pred_keys"] = np.array(sorted(pred.keys()))
    # boat pair (C1): PIL decode, grey, 2x box down-sample, crop to a multiple of 8
    from PIL import Image

    def boat(name):
        im = Image.open(os.path.join(REF, "assets", name)).convert("L")
        im = im.resize((im.width // 2, im.height // 2), Image.BOX)
        a = np.asarray(im, dtype=np.uint8)
        return a[: a.shape[0] // 8 * 8, : a.shape[1] // 8 * 8]

    b0, b1 = boat("boat1.png"), boat("boat2.png")
    out["boat_image0"], out["boat_image1"] = b0, b1
    t0 = torch.from_numpy(b0.astype(np.float32) / 255)[None, None]
    t1 = torch.from_numpy(b1.astype(np.float32) / 255)[None, None]
    s0 = torch.tensor([[float(b0.shape[1]), float(b0.shape[0])]])
    s1 = torch.tensor([[float(b1.shape[1]), float(b1.shape[0])]])
    pred = pipe({"view0": {"image": t0, "image_size": s0}, "view1": {"image": t1, "image_size": s1}})
    for key in keys:
        if key.startswith("descriptors"):
            continue
        out["boat_" + key] = npy(pred[key])
    save("pipeline", **out)


# --------------------------------------------------- C3: superpoint+lightglue-official, HPatches-shaped
from c3_inputs import C3_PAIRS, C3_RAGGED_PAIRS, C3_RAGGED_THRESHOLD, c3_pair  # noqa: E402  (shared with the GPU test)


def _official_pipeline(detection_threshold):
    """The reference's TwoViewPipeline in the superpoint+lightglue-official configuration (the official extractor's
    constructor fetch redirected to the generated weights; the in-tree matcher stands in for `lightglue_pretrained`,
    which needs the third-party package: same architecture, same checkpoints)."""
    sd = weights.superpoint_state_dict(0)
    orig = torch.hub.load_state_dict_from_url
    torch.hub.load_state_dict_from_url = lambda *a, **k: sd
    try:
        pipe = TwoViewPipeline({"extractor": {"name": "gluefactory_nonfree.superpoint", "max_num_keypoints": 1024,
                                              "detection_threshold": detection_threshold, "nms_radius": 3},
                                "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1, "flash": False,
                                            "depth_confidence": -1, "width_confidence": -1}}).eval()
    finally:
        torch.hub.load_state_dict_from_url = orig
    pipe.matcher.load_state_dict(weights.lightglue_state_dict(0), strict=False)
    return pipe


def _official_records(pipe, pairs):
    """One record per pair exactly as utils/export_predictions.py:36-85 writes it: export keys of
    eval/hpatches.py:61-68 (+ the optional score keys), key points multiplied by 1 / scales (:55-61)."""
    export_keys = ["keypoints0", "keypoints1", "matches0", "matches1", "matching_scores0", "matching_scores1",
                   "keypoint_scores0", "keypoint_scores1"]
    out = {"names": np.array([n for n, *_ in pairs])}
    counts = []
    for i, (name, seed, s0, s1, origs) in enumerate(pairs):
        data = c3_pair(seed, s0, s1, origs)
        pred = pipe(data)
        rec = {k: pred[k] for k in export_keys}
        for k in ("keypoints0", "keypoints1"):  # export_predictions.py:55-61
            rec[k] = rec[k] * (1.0 / data["view" + k[-1]]["scales"])[None]
        for k, v in rec.items():
            out[f"p{i}_{k}"] = npy(v[0])
        counts.append((rec["keypoints0"].shape[1], rec["keypoints1"].shape[1]))
        print(name, "kpts", *counts[-1], "matches", int((rec["matches0"] >= 0).sum()))
    return out, counts


def golden_pipeline_official():
    """BASELINE config 3 (superpoint+lightglue-official.yaml:3-13,27-33: gluefactory_nonfree.superpoint, 1024 key
    points, threshold 0, NMS 3 + LightGlue, filter 0.1) through the reference's TwoViewPipeline, b = 1."""
    out, _ = _official_records(_official_pipeline(0.0), C3_PAIRS)
    save("pipeline_official", **out)


def golden_pipeline_official_ragged():
    """The same configuration with `detection_threshold` raised so that views keep FEWER than 1024 key points and the
    two views of a pair keep different numbers: the official extractor's ragged path at config-3 sizes."""
    out, counts = _official_records(_official_pipeline(C3_RAGGED_THRESHOLD), C3_RAGGED_PAIRS)
    assert any(a != b and max(a, b) < 1024 for a, b in counts), counts  # a pair with two different counts below the cap
    assert any(max(a, b) == 1024 for a, b in counts) and any(min(a, b) < 1024 for a, b in counts), counts
    save("pipeline_official_ragged", **out)


def golden_boat_native():
    """C1 at its stated size (tests/test_integration.py:31-44,75-81): assets/boat1.png <-> boat2.png, 850 x 680 RGB,
    no resize, SuperPoint-open + LightGlue through the reference's TwoViewPipeline.  The images themselves are the
    reference's assets (data): stored as uint8 RGB."""
    from PIL import Image

    conf = {"extractor": {"name": "extractors.superpoint_open", "weights": SPO_PATH, "max_num_keypoints": 1024,
                          "detection_threshold": 0.0, "nms_radius": 3},
            "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1, "flash": False}}
    pipe = TwoViewPipeline(conf).eval()
    pipe.matcher.load_state_dict(weights.lightglue_state_dict(0), strict=False)
    out = {}
    views = {}
    for i, name in enumerate(("boat1.png", "boat2.png")):
        a = np.asarray(Image.open(os.path.join(REF, "assets", name)).convert("RGB"), dtype=np.uint8)
        out[f"image{i}"] = a  # [680, 850, 3]
        t = torch.from_numpy(a.astype(np.float32) / 255).permute(2, 0, 1)[None].contiguous()
        views[f"view{i}"] = {"image": t, "image_size": torch.tensor([[float(a.shape[1]), float(a.shape[0])]])}
    pred = pipe(views)
    for key in ("keypoints0", "keypoints1", "keypoint_scores0", "keypoint_scores1", "matches0", "matches1",
                "matching_scores0", "matching_scores1"):
        out[key] = npy(pred[key])
    print("boat native", out["image0"].shape, "matches", int((pred["matches0"] >= 0).sum()))
    # the same pair with filter_threshold 0 (every mutual arg-max pair is a match): with name-seeded weights no score of
    # this pair reaches 0.1, so the run above pins key points and scores but an EMPTY match set
    pipe0 = TwoViewPipeline({**conf, "matcher": {**conf["matcher"], "filter_threshold": 0.0}}).eval()
    pipe0.matcher.load_state_dict(weights.lightglue_state_dict(0), strict=False)
    pred0 = pipe0(views)
    for key in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
        out["th0_" + key] = npy(pred0[key])
    print("boat native, filter_threshold 0: matches", int((pred0["matches0"] >= 0).sum()))
    save("boat_native", **out)


def golden_lightglue_adaptive():
    """Adaptive width (point pruning) and the depth check on a run that reaches the last layer
    (lightglue.py:500-536,555-580); b == 1.  512 key points per image and a pruning rate of ~7 % per layer, so that
    the index scatter back to the un-pruned numbering is exercised on a few hundred surviving points and matches."""
    h, w, k = 240, 320, 512
    v0, v1, p0, p1 = _features(1, h, w, k, seed=31)
    size = torch.tensor([[w, h]], dtype=torch.float32)
    data = {"keypoints0": p0["keypoints"], "keypoints1": p1["keypoints"], "descriptors0": p0["descriptors"],
            "descriptors1": p1["descriptors"], "view0": {"image_size": size}, "view1": {"image_size": size}}
    out = {"keypoints0": npy(p0["keypoints"]), "keypoints1": npy(p1["keypoints"]),
           "descriptors0": npy(p0["descriptors"]), "descriptors1": npy(p1["descriptors"]), "image_size": npy(size)}
    sd = weights.lightglue_adaptive_state_dict(0, prune_z=ADAPTIVE_PRUNE_Z)
    for tag, conf in (("prune", {"width_confidence": 0.95}),
                      ("both", {"width_confidence": 0.95, "depth_confidence": 0.95})):
        m = ref_lg.LightGlue({"filter_threshold": 0.1, **conf}).eval()
        m.load_state_dict(sd, strict=False)
        pred = m(data)
        for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1",
                    "log_assignment"):
            out[f"{tag}_{key}"] = npy(pred[key])
        print(tag, "kept", pred["log_assignment"].shape, "matches", int((pred["matches0"] >= 0).sum()),
              "prune0 hist", torch.bincount(pred["prune0"][0].long()).tolist())
    save("lightglue_adaptive", **out)


ADAPTIVE_PRUNE_Z = 1.5  # tests/test_gpu_models.py builds the same weights


def golden_nn_matcher():
    """NearestNeighborMatcher (matchers/nearest_neighbor_matcher.py) on SuperPoint descriptors."""
    from gluefactory.models.matchers.nearest_neighbor_matcher import NearestNeighborMatcher

    _, _, p0, p1 = _features(2, 120, 160, 160, seed=31)
    d0, d1 = p0["descriptors"], p1["descriptors"][:, :140]
    out = {"descriptors0": npy(d0), "descriptors1": npy(d1)}
    for tag, conf in (("default", {}), ("ratio", {"ratio_thresh": 0.9}), ("dist", {"distance_thresh": 0.9}),
                      ("nomutual", {"mutual_check": False, "ratio_thresh": 0.95, "distance_thresh": 1.1})):
        pred = NearestNeighborMatcher(conf).eval()({"descriptors0": d0, "descriptors1": d1})
        for key in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
            out[f"{tag}_{key}"] = npy(pred[key])
        if tag == "default":
            out["similarity"] = npy(pred["similarity"])
            out["log_assignment"] = npy(pred["log_assignment"])
        print(tag, int((pred["matches0"] >= 0).sum()))
    save("nn_matcher", **out)


def golden_homography():
    """Reference geometry used by the HPatches match metrics (gluefactory/geometry/homography.py:161-180,314-344)."""
    from gluefactory.geometry import homography as ref_h

    g = torch.Generator().manual_seed(17)
    H = torch.eye(3)[None].repeat(3, 1, 1)
    H[:, :2, :2] += 0.2 * torch.randn((3, 2, 2), generator=g)
    H[:, :2, 2] = 30 * torch.randn((3, 2), generator=g)
    H[:, 2, :2] = 1e-4 * torch.randn((3, 2), generator=g)
    pts = torch.rand((3, 50, 2), generator=g) * torch.tensor([640.0, 480.0])
    out = {"H": npy(H), "pts": npy(pts)}
    out["warp_fwd"] = npy(ref_h.warp_points_torch(pts, H, inverse=False))
    out["warp_inv"] = npy(ref_h.warp_points_torch(pts, H, inverse=True))
    noisy = ref_h.warp_points_torch(pts, H, inverse=False) + 2 * torch.randn((3, 50, 2), generator=g)
    out["pts1"] = npy(noisy)
    out["sym_err"] = npy(torch.stack([ref_h.sym_homography_error(pts[i], noisy[i], H[i]) for i in range(3)]))
    H2 = H.clone()
    H2[:, :2, 2] += 1.5
    out["corner_err"] = npy(torch.stack([ref_h.homography_corner_error(H2[i], H[i], torch.tensor([640.0, 480.0]))
                                         for i in range(3)]))
    save("homography", **out)


def golden_scale_ori():
    """LightGlue with add_scale_ori (SIFT-style inputs: scales0/1, oris0/1 feed the positional encoding)."""
    sd = weights.lightglue_state_dict(0, add_scale_ori=True)
    m = ref_lg.LightGlue({"weights": None, "filter_threshold": 0.1, "add_scale_ori": True, "flash": False}).eval()
    missing = m.load_state_dict(sd, strict=False)
    assert missing.missing_keys == ["confidence_thresholds"] and not missing.unexpected_keys, missing
    _, _, p0, p1 = _features(2, 120, 160, 96, 41)
    kp0, d0 = p0["keypoints"], p0["descriptors"]
    kp1, d1 = p1["keypoints"][:, :80], p1["descriptors"][:, :80]
    g = torch.Generator().manual_seed(9)
    sc0, sc1 = torch.rand((2, 96), generator=g) * 4 + 1, torch.rand((2, 80), generator=g) * 4 + 1
    or0 = (torch.rand((2, 96, 1), generator=g) - 0.5) * 6.28   # one side [B,K,1], the other [B,K]: both accepted
    or1 = (torch.rand((2, 80), generator=g) - 0.5) * 6.28
    size = torch.tensor([[160.0, 120.0]] * 2)
    out = m({"keypoints0": kp0, "keypoints1": kp1, "descriptors0": d0, "descriptors1": d1, "scales0": sc0, "scales1": sc1,
             "oris0": or0, "oris1": or1, "view0": {"image_size": size}, "view1": {"image_size": size}})
    save("scale_ori", keypoints0=npy(kp0), keypoints1=npy(kp1), descriptors0=npy(d0), descriptors1=npy(d1),
         scales0=npy(sc0), scales1=npy(sc1), oris0=npy(or0), oris1=npy(or1), image_size=npy(size),
         matches0=npy(out["matches0"]), matches1=npy(out["matches1"]), matching_scores0=npy(out["matching_scores0"]),
         log_assignment=npy(out["log_assignment"]), ref_descriptors0=npy(out["ref_descriptors0"]))


# ------------------------------------------- the benchmarked configuration itself (BASELINE configs[1] and [3])
from bench_inputs import C4_PAIRS, DESC_SAMPLE_ROWS, FP64_DEPTH, FP64_IMAGES, desc_checksum_weights  # noqa: E402


def _bench_pipeline(k):
    conf = {"extractor": {"name": "extractors.superpoint_open", "weights": SPO_PATH, "max_num_keypoints": k,
                          "detection_threshold": 0.0, "nms_radius": 3},
            "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1, "flash": False,
                        "depth_confidence": -1, "width_confidence": -1}}
    pipe = TwoViewPipeline(conf).eval()
    pipe.matcher.load_state_dict(weights.lightglue_state_dict(0), strict=False)
    return pipe


def _bench_records(pipe, v0, v1, pair_ids, k):
    """The reference's TwoViewPipeline, batch 1 per pair as the evaluation calls it (two_view_pipeline.py:278-339,
    export_predictions.py:36-52), on pairs of the benchmark's own synthetic batch."""
    import hashlib
    import time

    h, w = v0.shape[-2:]
    size = torch.tensor([[float(w), float(h)]])
    wts = desc_checksum_weights()
    rows = torch.linspace(0, k - 1, DESC_SAMPLE_ROWS).round().long()
    n = len(pair_ids)
    out = {"pair_ids": np.asarray(pair_ids, np.int32), "desc_rows": rows.numpy().astype(np.int32),
           "keypoints": np.zeros((n, 2, k, 2), np.int16), "keypoint_scores": np.zeros((n, 2, k), np.float32),
           "matches": np.zeros((n, 2, k), np.int32), "matching_scores": np.zeros((n, 2, k), np.float32),
           "desc_sample": np.zeros((n, 2, DESC_SAMPLE_ROWS, 256), np.float32),
           "desc_checksum": np.zeros((n, 2, 2, k), np.float32), "image_sha256": []}
    t0 = time.time()
    for j, i in enumerate(pair_ids):
        pred = pipe({"view0": {"image": v0[i:i + 1], "image_size": size}, "view1": {"image": v1[i:i + 1], "image_size": size}})
        for s in (0, 1):
            kp = pred[f"keypoints{s}"][0]
            assert kp.shape == (k, 2), kp.shape  # every image of these workloads has more than k detections
            out["keypoints"][j, s] = (kp - 0.5).round().short().numpy()  # integer pixel (x, y); key point = pixel + 0.5
            assert torch.equal(torch.from_numpy(out["keypoints"][j, s]).float() + 0.5, kp)
            out["keypoint_scores"][j, s] = npy(pred[f"keypoint_scores{s}"][0])
            out["matches"][j, s] = npy(pred[f"matches{s}"][0]).astype(np.int32)
            out["matching_scores"][j, s] = npy(pred[f"matching_scores{s}"][0])
            d = pred[f"descriptors{s}"][0]
            out["desc_sample"][j, s] = npy(d[rows])
            out["desc_checksum"][j, s] = npy(wts @ d.T)
        out["image_sha256"].append(hashlib.sha256(npy(v0[i]).tobytes() + npy(v1[i]).tobytes()).hexdigest())
        print(f"  pair {i}: {int((pred['matches0'] >= 0).sum())} matches, {time.time() - t0:.0f} s", flush=True)
    out["image_sha256"] = np.array(out["image_sha256"])
    return out


def golden_c2_batch32():
    """BASELINE configs[1] at the batch bench.py times: all 32 pairs of `synthetic_pairs(32, 480, 640, seed=1234)`
    (bench.py rank 0), 1024 key points, through the reference itself."""
    v0, v1 = synthetic.synthetic_pairs(32, 480, 640, seed=1234)
    save("c2_batch32", **_bench_records(_bench_pipeline(1024), v0, v1, list(range(32)), 1024))


def golden_c4_pairs():
    """BASELINE configs[3]: four pairs of `synthetic_pairs(32, 1024, 1024, seed=1234)` (bench.py --workload c4, rank 0),
    2048 key points, through the reference itself."""
    v0, v1 = synthetic.synthetic_pairs(32, 1024, 1024, seed=1234)
    save("c4_pairs", **_bench_records(_bench_pipeline(2048), v0, v1, list(C4_PAIRS), 2048))


def golden_c2_fp64_order():
    """What the top-1024 order of the C2 images is in exact arithmetic: the reference's SuperPoint-open module cast to
    float64 (same fp32 weights, same fp32 images, every product and sum in double) on views 0/1 of pairs 0..7.  Used by
    tools/order_exactness.py to count how many rank swaps each fp32 evaluation (torch-CPU = the reference, and the HIP
    convolution variants) has against that order (superpoint_open.py:54-58,156-192)."""
    v0, v1 = synthetic.synthetic_pairs(32, 480, 640, seed=1234)
    m = make_spo(max_num_keypoints=FP64_DEPTH, detection_threshold=0.0, nms_radius=3).double()
    n = FP64_IMAGES // 2
    kpts = np.zeros((n, 2, FP64_DEPTH, 2), np.int16)
    scores = np.zeros((n, 2, FP64_DEPTH), np.float64)
    for i in range(n):
        for s, v in ((0, v0), (1, v1)):
            # the module's own forward casts the key points to float32 before grid_sample (superpoint_open.py:176), so
            # the detection half is driven through its sub-modules and functions, as `_forward` does (:132-192)
            logits = m.detector(m.backbone(v[i:i + 1].double()))
            assert logits.dtype == torch.float64
            heat = torch.softmax(logits, 1)[:, :-1]
            b, _, h, w = heat.shape
            heat = heat.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
            heat = ref_spo.batched_nms(heat, 3)
            heat[:, :4] = -1
            heat[:, :, :4] = -1
            heat[:, -4:] = -1
            heat[:, :, -4:] = -1
            heat = heat.squeeze(0)
            idxs = torch.where(heat > 0.0)
            k_all = torch.stack(idxs[-2:], dim=-1).flip(1)
            k, sc = ref_spo.select_top_k_keypoints(k_all, heat[idxs], FP64_DEPTH)
            assert k.shape == (FP64_DEPTH, 2) and sc.dtype == torch.float64
            kpts[i, s] = k.short().numpy()
            scores[i, s] = sc.numpy()
            print(f"  fp64 pair {i} view {s}: min gap in top-1024 {np.min(-np.diff(scores[i, s][:1024])):.3g}", flush=True)
    save("c2_fp64_order", keypoints=kpts, scores=scores)


PAD_SEED = 4242


def golden_pad_random_c():
    """`force_num_keypoints` with images that keep FEWER than k key points: `pad_and_stack(mode="random_c")`
    (models/utils/misc.py:48-60,103-113; superpoint_open.py:193-207) draws the padding from torch's CPU generator.  The
    generator is seeded with PAD_SEED right before each call; the HIP module's opt-in `pad_random: "torch_cpu"` must
    then reproduce the padded key points bit for bit (tests/test_gpu_models.py)."""
    img = synthetic.synthetic_images(3, 120, 160, seed=51)
    out = {"image": npy(img), "seed": np.array(PAD_SEED)}
    m = make_spo(max_num_keypoints=680, detection_threshold=0.0, nms_radius=3, force_num_keypoints=True)
    counts = [int(make_spo(max_num_keypoints=4096, detection_threshold=0.0, nms_radius=3)({"image": img[i:i + 1]})
                  ["keypoints"].shape[1]) for i in range(3)]
    torch.manual_seed(PAD_SEED)
    p = m({"image": img})
    out["mixed_counts"] = np.array(counts)
    assert min(counts) < 680 < max(counts), counts  # padded and un-padded images in one batch
    for key in ("keypoints", "keypoint_scores"):
        out["mixed_" + key] = npy(p[key])
    out["mixed_descriptors_tail"] = npy(p["descriptors"][:, -48:])  # the padded rows (at most 13) and real rows before them
    # with image_size (the bounds of an image WITHOUT key points) and a threshold nothing passes: all padding
    size = torch.tensor([[150.0, 110.0]] * 3)
    m0 = make_spo(max_num_keypoints=32, detection_threshold=2.0, nms_radius=3, force_num_keypoints=True)
    torch.manual_seed(PAD_SEED)
    p = m0({"image": img, "image_size": size})
    out["empty_image_size"] = npy(size)
    for key in ("keypoints", "keypoint_scores", "descriptors"):
        out["empty_" + key] = npy(p[key])
    print("pad_random_c: candidates per image", counts, "of 680; empty case keypoints in",
          float(p["keypoints"].min()), float(p["keypoints"].max()))
    save("pad_random_c", **out)


if __name__ == "__main__":
    if "--only-pad" in sys.argv:
        golden_pad_random_c()
        sys.exit(0)
    if "--only-c2-batch32" in sys.argv:
        golden_c2_batch32()
        sys.exit(0)
    if "--only-c4-pairs" in sys.argv:
        golden_c4_pairs()
        sys.exit(0)
    if "--only-c2-fp64" in sys.argv:
        golden_c2_fp64_order()
        sys.exit(0)
    if "--only-official-pipeline" in sys.argv:
        golden_pipeline_official()
        sys.exit(0)
    if "--only-official-ragged" in sys.argv:
        golden_pipeline_official_ragged()
        sys.exit(0)
    if "--only-boat-native" in sys.argv:
        golden_boat_native()
        sys.exit(0)
    if "--only-adaptive" in sys.argv:
        golden_lightglue_adaptive()
        sys.exit(0)
    if "--only-scale-ori" in sys.argv:
        golden_scale_ori()
        sys.exit(0)
    if "--only-specular" in sys.argv:
        golden_specular()
        sys.exit(0)
    golden_specular()
    golden_scale_ori()
    golden_homography()
    golden_lightglue_adaptive()
    golden_nn_matcher()
    golden_nms()
    golden_assignment()
    golden_superpoint_open()
    golden_superpoint_official()
    golden_lightglue()
    golden_pipeline()
    golden_pipeline_official()
    golden_pipeline_official_ragged()
    golden_boat_native()
    golden_c2_batch32()
    golden_c4_pairs()
    golden_c2_fp64_order()
    golden_pad_random_c()
