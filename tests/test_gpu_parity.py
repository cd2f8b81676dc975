"""-m gpu: the HIP path (through libsbbseg's C ABI) against the CPU oracle on the same seeded inputs,
against the golden fixtures captured from the reference loop, and -- at BASELINE sizes -- through
size-independent properties.  Forward tolerances are stated in tests/gpu_common.py."""
import json
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gpu_common import (EXACT_MARGIN, TOL_LABEL_FRAC_F16, TOL_LAYER_REL, TOL_SOFTMAX, compare_probs, exact_label_check,  # noqa: E402
                        make_model, patches_from_page)
from oracle import keras_forward as kf  # noqa: E402
from oracle import tiling  # noqa: E402
from sbb_textline_detection_amd import _capi, predict  # noqa: E402
from sbb_textline_detection_amd.synthetic import noise_page, synthetic_page  # noqa: E402

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tiling_golden.json")))["cases"]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


# ------------------------------------------------------------------------------------------ ingest
@pytest.mark.parametrize("precision", ["f16", "bf16", "f32", "f16x3"])
def test_ingest_forms(precision):
    cfg, w, g, model = make_model(2, 64, 96, precision=precision, calib_hw=64)
    page = noise_page(200, 300, 1)
    xy = np.array([[0, 0], [204, 136], [17, 55]], np.int32)
    c8 = model.ctx.debug_ingest(page, xy, _capi.INPUT_C8, (64, 96, 8))
    pairs = model.ctx.debug_ingest(page, xy, _capi.INPUT_PAIRS, (64 + 6, (96 + 6 + 1) // 2, 8))
    import torch
    for k, (x0, y0) in enumerate(xy):
        ref = (page[y0:y0 + 64, x0:x0 + 96] / 255.0).astype(np.float32)       # main.py:239, 285
        if precision in ("f16", "bf16"):
            ref = torch.from_numpy(ref).to(torch.bfloat16 if precision == "bf16" else torch.float16).to(torch.float32).numpy()
        padded = np.zeros((70, 102, 4), np.float32)
        padded[3:67, 3:99, :3] = ref
        if precision == "f16x3":          # hi + lo halves carry f32(v/255) to ~22 bits (read back as their fp32 sum)
            assert np.allclose(c8[k, :, :, :3], ref, rtol=2.0 ** -21, atol=0) and not c8[k, :, :, 3:].any()
            assert np.allclose(pairs[k], padded.reshape(70, 51, 8), rtol=2.0 ** -21, atol=0)
            continue
        assert np.array_equal(c8[k, :, :, :3], ref) and not c8[k, :, :, 3:].any()
        assert np.array_equal(pairs[k], padded.reshape(70, 51, 8))
    model.release()


# --------------------------------------------------------------------------------- forward, by layer
@pytest.mark.parametrize("precision", ["f32", "f16x3", "f16", "bf16"])
def test_every_fused_layer_matches_oracle(precision, monkeypatch):
    monkeypatch.setenv("SBBSEG_FUSE_BLOCKS", "0")      # this test reads every intermediate tensor: keep the bottleneck blocks as three convs
    cfg, w, g, model = make_model(2, 64, 96, seed=2, precision=precision, max_batch=4, calib_hw=64)
    # (calibrating BN on 64x64 crops leaves stage 5 with 8 samples per channel: activations reach
    #  ~1e3 and the softmax of this tiny net is ill-conditioned -- it is only asserted for f32 here;
    #  the 16-bit softmax tolerance is checked on the properly calibrated 448 nets below)
    x8 = patches_from_page(64, 96, 3, seed=4)
    x = (x8 / 255.0).astype(np.float32)
    taps = {name: None for name in model.plan.layer_tensor}
    ref = kf.forward(g, w, x, taps=taps)
    got = model.predict(x)
    rel_tol = TOL_LAYER_REL[precision]
    worst = ("", 0.0)
    for name, tid in model.plan.layer_tensor.items():
        t = model.plan.tensors[tid]
        a = model.ctx.debug_read_tensor(tid, 3, (t.H, t.W, t.C))
        r = taps[name]
        rel = float(np.abs(a - r).max() / (np.abs(r).max() + 1e-6))
        if rel > worst[1]:
            worst = (name, rel)
        assert rel < rel_tol, f"{name}: rel err {rel:.4g} ({precision})"
    d, mism, bad = compare_probs(ref, got, TOL_SOFTMAX.get(precision, 1.0))
    print(f"[layers {precision}] worst layer {worst}, max|dsoftmax| {d:.4f}, label mismatches {mism}")
    if precision in ("f32", "f16x3"):
        assert d < TOL_SOFTMAX[precision] and bad == 0, (d, mism, bad, worst)
    model.release()


@pytest.mark.parametrize("precision,hw", [("f16", (64, 96)), ("bf16", (64, 96)), ("f16", (224, 256))])
def test_fused_bottleneck_blocks_match_their_three_convs(precision, hw):
    """sbbseg_finalize folds each stage-2 bottleneck block (1x1 -> 3x3 -> 1x1 + shortcut) into one bottleneck_fused launch.
    Same network, fused vs the three convs it replaces (conv variant bit 18), every tensor the fused plan still writes,
    and the block outputs against the fp32 oracle (main.py:225-380 runs whatever Keras graph the .h5 holds: Add/ReLU nodes
    of keras_graph)."""
    h, wd = hw
    cfg, w, g, model = make_model(2, h, wd, seed=5, precision=precision, max_batch=4, calib_hw=min(160, max(h, wd)))
    names = [o["name"] for o in model.ctx.ops()]
    blocks = [n for n in names if n.startswith("block")]
    assert len(blocks) == 3 and sum("proj" in n for n in blocks) == 1, names
    x = (patches_from_page(h, wd, 3, seed=8) / 255.0).astype(np.float32)
    taps = {name: None for name in model.plan.layer_tensor}
    ref = kf.forward(g, w, x, taps=taps)
    stage2 = None

    def read_all():
        out = {}
        for name, tid in model.plan.layer_tensor.items():
            t = model.plan.tensors[tid]
            out[name] = model.ctx.debug_read_tensor(tid, 3, (t.H, t.W, t.C))
        return out
    got_fused = model.predict(x)
    t_fused = read_all()
    model.ctx.set_conv_variant(1 << 18)
    got_parts = model.predict(x)
    t_parts = read_all()
    model.ctx.set_conv_variant(1 << 20)                  # the one-group form of the fused kernel (default: producer / consumer): same bytes
    got_pq = model.predict(x)
    t_pq = read_all()
    model.ctx.set_conv_variant(0)
    assert np.array_equal(got_pq, got_fused)
    for name, tid in model.plan.layer_tensor.items():
        t = model.plan.tensors[tid]
        if not (t.C == 64 and t.H == max(model.plan.tensors[k].H for k in model.plan.layer_tensor.values() if model.plan.tensors[k].C == 256)):
            assert np.array_equal(t_pq[name], t_fused[name]), name
    stage2 = max(model.plan.tensors[tid].H for tid in model.plan.layer_tensor.values() if model.plan.tensors[tid].C == 256)
    checked = 0
    for name, tid in model.plan.layer_tensor.items():
        t = model.plan.tensors[tid]
        if t.C == 64 and t.H == stage2:
            continue                                   # the blocks' internal 64-channel tensors are not written by the fused plan
        a, b = t_fused[name], t_parts[name]
        scale = np.abs(b).max() + 1e-6
        assert np.abs(a - b).max() / scale < (4e-3 if precision == "f16" else 3e-2), (name, np.abs(a - b).max() / scale)
        if t.C == 256 and t.H == stage2:
            r = taps[name]
            rel = float(np.abs(a - r).max() / (np.abs(r).max() + 1e-6))
            assert rel < TOL_LAYER_REL[precision], (name, rel)
            checked += 1
    assert checked >= 3
    d = float(np.abs(got_fused - got_parts).max())
    print(f"[fused blocks {precision} {h}x{wd}] max|dsoftmax| fused vs three convs {d:.2e}; vs oracle {float(np.abs(got_fused - ref).max()):.2e}")
    assert d < (0.02 if precision == "f16" else 0.15)
    model.release()


@pytest.mark.parametrize("hw", [(64, 96), (224, 256)])
def test_fused_x3_identity_blocks_match_their_three_convs(hw):
    """Split mode: the three blocks of stage 2 (projection block + two identity blocks) run as one block_x3 launch each
    (csrc/block_x3.hip).  Fused vs unfused (conv variant bit 18) on every tensor the fused plan still writes: the same MFMA
    K order and the same epilogue formulas -> bit-identical; block outputs against the fp32 oracle within the layer tolerance."""
    h, wd = hw
    cfg, w, g, model = make_model(2, h, wd, seed=5, precision="f16x3", max_batch=4, calib_hw=min(160, max(h, wd)))
    names = [o["name"] for o in model.ctx.ops()]
    blocks = [n for n in names if n.startswith("block")]
    assert len(blocks) == 3 and sum("proj" in n for n in blocks) == 1, names
    x = (patches_from_page(h, wd, 3, seed=8) / 255.0).astype(np.float32)
    taps = {name: None for name in model.plan.layer_tensor}
    ref = kf.forward(g, w, x, taps=taps)

    def read_all():
        out = {}
        for name, tid in model.plan.layer_tensor.items():
            t = model.plan.tensors[tid]
            out[name] = model.ctx.debug_read_tensor(tid, 3, (t.H, t.W, t.C))
        return out
    got_fused = model.predict(x)
    t_fused = read_all()
    model.ctx.set_conv_variant(1 << 18)
    got_parts = model.predict(x)
    t_parts = read_all()
    model.ctx.set_conv_variant(0)
    stage2 = max(model.plan.tensors[tid].H for tid in model.plan.layer_tensor.values() if model.plan.tensors[tid].C == 256)
    checked = 0
    for name, tid in model.plan.layer_tensor.items():
        t = model.plan.tensors[tid]
        if t.C == 256 and t.H == stage2:
            a, b, r = t_fused[name], t_parts[name], taps[name]
            assert np.array_equal(a, b), (name, float(np.abs(a - b).max()))
            rel = float(np.abs(a - r).max() / (np.abs(r).max() + 1e-6))
            assert rel < TOL_LAYER_REL["f16x3"], (name, rel)
            checked += 1
    assert checked >= 3
    assert np.array_equal(got_fused, got_parts)
    d = float(np.abs(got_fused - ref).max())
    print(f"[fused x3 blocks {h}x{wd}] max|dsoftmax| vs oracle {d:.2e}")
    assert d < TOL_SOFTMAX["f16x3"] and exact_label_check(ref, got_fused)[1] == 0
    model.release()


@pytest.mark.parametrize("precision", ["f16x3", "f16"])
@pytest.mark.parametrize("hw,nb", [((64, 96), 3), ((224, 256), 5), ((448, 448), 9)])
def test_fused_x3_stem_pool_matches_the_two_launches(hw, nb, precision):
    """Split mode (and the plain fp16 mode: `stem_pool<false>`, one plane, the pool reads back the ROUNDED f1): the stem and its max-pool run as ONE launch (csrc/stem_pool_x3.hip: pool windows taken from the epilogue's values,
    channel-half blocks walking 16-row strips, one recomputed row per tile).  Against the two-launch form (conv variant bit 22:
    stem_conv_pairs_x3 + maxpool_kernel) every tensor of the plan -- the stem's raw f1 skip, the pooled tensor, everything after --
    must be the same bits: the pool is the maximum over the same fp32 numbers (hi + lo of the stored f1, one fma, ReLU)."""
    h, wd = hw
    os.environ["SBBSEG_STEM_POOL_F16"] = "1"                 # (the plain mode's fused form is opt-in: no faster than the two launches)
    try:
        cfg, w, g, model = make_model(2, h, wd, seed=6, precision=precision, max_batch=nb + 2, calib_hw=min(160, max(h, wd)))
    finally:
        del os.environ["SBBSEG_STEM_POOL_F16"]
    names = [o["name"] for o in model.ctx.ops()]
    assert any(n.startswith("stem_") for n in names) and any(n.startswith("maxpool") for n in names), names
    x = (patches_from_page(h, wd, nb, seed=18) / 255.0).astype(np.float32)

    def read_all():
        out = {}
        for name, tid in model.plan.layer_tensor.items():
            t = model.plan.tensors[tid]
            out[name] = model.ctx.debug_read_tensor(tid, nb, (t.H, t.W, t.C))
        return out
    got_fused = model.predict(x)
    t_fused = read_all()
    model.ctx.set_conv_variant(1 << 22)
    got_two = model.predict(x)
    t_two = read_all()
    model.ctx.set_conv_variant(0)
    assert np.array_equal(model.predict(x), got_fused)                       # and back again: deterministic
    pooled = [n for n, tid in model.plan.layer_tensor.items() if model.plan.tensors[tid].C == 64 and model.plan.tensors[tid].H == h // 4 - 1]
    assert pooled, list(model.plan.layer_tensor)
    for name in t_fused:
        assert t_fused[name].shape == t_two[name].shape
        assert np.array_equal(t_fused[name], t_two[name]), (name, float(np.abs(t_fused[name] - t_two[name]).max()))
    assert float(np.abs(t_fused[pooled[0]]).max()) > 0
    assert np.array_equal(got_fused, got_two)
    ref = kf.forward(g, w, x[:2])
    if precision == "f16x3":
        assert float(np.abs(got_fused[:2] - ref).max()) < TOL_SOFTMAX["f16x3"] and exact_label_check(ref, got_fused[:2])[1] == 0
    else:
        assert float(np.abs(got_fused[:2] - ref).max()) < 0.2
    model.release()


@pytest.mark.parametrize("precision", ["f16x3", "f16"])
@pytest.mark.parametrize("hw,nb", [((64, 96), 3), ((224, 256), 5), ((448, 448), 9)])
def test_dec_halo_x3_matches_the_generic_kernel(hw, nb, precision):
    """Split mode (and, dec_halo_f16.hip, the plain fp16 mode: 17 K-steps of two k-halves, both halos double-buffered, pair-row skip
    layout): the decoder conv at full / 2 resolution (four parity classes of conv3x3([up2(128 ch), skip 64 ch]) -> 64 ch) runs
    dec_halo_x3 (csrc/dec_halo_x3.hip: 16 x 16 output tiles, source halos resident in LDS, weights streamed as A fragments, every
    vector load counted by hand).  Against conv_igemm_mfma's grouped launch (conv variant bit 23) every tensor of the plan must be the
    same bits: same K-steps in the same order, same three MFMAs per product, same epilogue.  Run three times: the hand-placed waits
    (vmcnt counts that include the halo DMA bursts and the epilogue's stores) must not race."""
    h, wd = hw
    cfg, w, g, model = make_model(2, h, wd, seed=8, precision=precision, max_batch=nb + 2, calib_hw=min(160, max(h, wd)))
    x = (patches_from_page(h, wd, nb, seed=28) / 255.0).astype(np.float32)

    def read_all():
        out = {}
        for name, tid in model.plan.layer_tensor.items():
            t = model.plan.tensors[tid]
            out[name] = model.ctx.debug_read_tensor(tid, nb, (t.H, t.W, t.C))
        return out
    got_halo = model.predict(x)
    t_halo = read_all()
    model.ctx.set_conv_variant(1 << 23)
    got_gen = model.predict(x)
    t_gen = read_all()
    model.ctx.set_conv_variant(0)
    dec4 = [n for n, tid in model.plan.layer_tensor.items() if model.plan.tensors[tid].C == 64 and model.plan.tensors[tid].H == h // 2]
    assert dec4, list(model.plan.layer_tensor)
    for name in t_halo:
        assert np.array_equal(t_halo[name], t_gen[name]), (name, float(np.abs(t_halo[name] - t_gen[name]).max()))
    assert np.array_equal(got_halo, got_gen)
    for _ in range(3):
        assert np.array_equal(model.predict(x), got_halo)
    ref = kf.forward(g, w, x[:2])
    if precision == "f16x3":
        assert float(np.abs(got_halo[:2] - ref).max()) < TOL_SOFTMAX["f16x3"] and exact_label_check(ref, got_halo[:2])[1] == 0
    else:
        assert float(np.abs(got_halo[:2] - ref).max()) < 0.2
    model.release()


@pytest.mark.parametrize("precision", ["f16x3", "f16"])
@pytest.mark.parametrize("hw,nb", [((64, 96), 3), ((224, 256), 5), ((448, 448), 9), ((448, 448), 2)])
def test_expand_reduce_x3_matches_the_two_launches(hw, nb, precision):
    """Split mode (and the plain fp16 mode: the same kernel on K-steps of two k-halves), encoder stages 3 / 4: an identity block's last
    1x1 conv (+ residual, ReLU) and the next block's first 1x1 conv run as ONE
    launch (csrc/expand_reduce_x3.hip: the 4C-channel tensor is written once and contracted from LDS, weights streamed, every vector-memory
    operation counted by hand).  Against the two conv_igemm_mfma launches (conv variant bit 24) every tensor of the plan must be the same
    bits.  64 x 96 patches give 3 x 96 / 3 x 24 pixels per launch: ragged last tiles (loads read zeros, stores are dropped past the end).
    Run three times: the hand-placed waits must not race."""
    h, wd = hw
    cfg, w, g, model = make_model(2, h, wd, seed=9, precision=precision, max_batch=nb + 2, calib_hw=min(160, max(h, wd)))
    x = (patches_from_page(h, wd, nb, seed=29) / 255.0).astype(np.float32)

    def read_all():
        out = {}
        for name, tid in model.plan.layer_tensor.items():
            t = model.plan.tensors[tid]
            out[name] = model.ctx.debug_read_tensor(tid, nb, (t.H, t.W, t.C))
        return out
    model.ctx.set_conv_variant(1 << 25)              # (the 3x3 conv in front of a stage-3 pair keeps its own launch here: conv3_expand_reduce has its own test)
    got_fused = model.predict(x)
    t_fused = read_all()
    model.ctx.set_conv_variant((1 << 24) | (1 << 25))
    got_two = model.predict(x)
    t_two = read_all()
    model.ctx.set_conv_variant(1 << 25)
    for name in t_fused:
        assert np.array_equal(t_fused[name], t_two[name]), (name, float(np.abs(t_fused[name] - t_two[name]).max()))
    assert np.array_equal(got_fused, got_two)
    for _ in range(3):
        assert np.array_equal(model.predict(x), got_fused)
    ref = kf.forward(g, w, x[:2])
    if precision == "f16x3":
        assert float(np.abs(got_fused[:2] - ref).max()) < TOL_SOFTMAX["f16x3"] and exact_label_check(ref, got_fused[:2])[1] == 0
    else:
        assert float(np.abs(got_fused[:2] - ref).max()) < 0.2
    model.release()


@pytest.mark.parametrize("precision", ["f16x3", "f16"])
@pytest.mark.parametrize("hw,nb", [((64, 64), 5), ((128, 192), 4), ((448, 448), 9), ((448, 448), 2)])
def test_conv3_expand_reduce_matches_the_three_launches(hw, nb, precision):
    """Encoder stage 3 (C = 128, maps whose sides are multiples of 8): an identity block's 3x3 conv, its last 1x1 conv (+ residual, ReLU) and
    the next block's first 1x1 conv run as ONE launch (csrc/conv3_expand_reduce.hip: the 3x3's output b lives in LDS only).  Against the
    3x3 conv's own launch in front of expand_reduce (conv variant bit 25) and against three conv_igemm_mfma launches (bits 24 + 25) every
    tensor the fused plan still writes must be the same bits; the tensors it no longer writes are exactly the two b's (activation buffers
    are filled with NaNs before the fused run).  64 x 64 patches: one 8 x 8 tile per patch, every halo pixel outside the image."""
    h, wd = hw
    cfg, w, g, model = make_model(2, h, wd, seed=11, precision=precision, max_batch=nb + 2, calib_hw=min(160, max(h, wd)))
    x = (patches_from_page(h, wd, nb, seed=31) / 255.0).astype(np.float32)

    def read_all():
        out = {}
        for name, tid in model.plan.layer_tensor.items():
            t = model.plan.tensors[tid]
            out[name] = model.ctx.debug_read_tensor(tid, nb, (t.H, t.W, t.C))
        return out
    model.ctx.poison_activations(0xFF)
    got_fused = model.predict(x)
    t_fused = read_all()
    model.ctx.set_conv_variant(1 << 25)
    got_pair = model.predict(x)
    t_pair = read_all()
    model.ctx.set_conv_variant((1 << 24) | (1 << 25))
    got_three = model.predict(x)
    t_three = read_all()
    model.ctx.set_conv_variant(0)
    skipped = []
    for name in t_fused:
        # (tensors inside the fused stage-2 blocks are written by none of the three runs: NaN everywhere, in all of them)
        assert np.array_equal(t_pair[name], t_three[name], equal_nan=True), name
        if np.isnan(t_fused[name]).all() and not np.isnan(t_three[name]).any():
            skipped.append(name)
            continue
        assert np.array_equal(t_fused[name], t_three[name], equal_nan=True), (name, float(np.nanmax(np.abs(t_fused[name] - t_three[name]))))
    shapes = {model.plan.tensors[model.plan.layer_tensor[n]].C for n in skipped}
    assert len(skipped) == 2 and shapes == {128}, skipped            # blocks 2 and 3 of stage 3: their 3x3 outputs never reach HBM
    assert np.array_equal(got_fused, got_three) and np.array_equal(got_pair, got_three)
    for _ in range(3):                                               # the hand-placed waits must not race
        assert np.array_equal(model.predict(x), got_fused)
    ref = kf.forward(g, w, x[:2])
    if precision == "f16x3" and hw == (448, 448):
        assert float(np.abs(got_fused[:2] - ref).max()) < TOL_SOFTMAX["f16x3"] and exact_label_check(ref, got_fused[:2])[1] == 0
    model.release()


@pytest.mark.parametrize("precision", ["f16", "f16x3"])
def test_conv_tile_families_agree(precision):
    """4-wave/2-stage and 8-wave/3-stage conv tiles, persistent or one block per tile, compute the same sums -- in the split mode
    too, where variants 1-3 send the stem to the generic kernel and the max-pool op must then launch itself (round-4 advisory:
    it returned early on `fused_into_stem` alone and left a stale pool tensor)."""
    cfg, w, g, model = make_model(2, 224, 224, seed=3, precision=precision, max_batch=6)
    x = (patches_from_page(224, 224, 5, seed=2) / 255.0).astype(np.float32)
    variants = (1, 2, 3, 5, 6, 8, 16, 32, 0x220, 64, 128, 0x10000, 0x10003, 0) if precision == "f16" else (1, 2, 3, 5, 8, 32, 0x10003, 0)
    # bit 2 = one block per tile instead of persistent blocks; bit 3 = no XCD grouping; 3 = big 8-wave tiles; 16 = half-K-step stages in a 4-deep ring
    other = (patches_from_page(224, 224, 5, seed=9) / 255.0).astype(np.float32)
    outs = []
    for variant in variants:
        model.ctx.set_conv_variant(0)
        model.predict(other)                         # every activation buffer now holds ANOTHER input's tensors: an op that wrongly
        model.ctx.set_conv_variant(variant)          # launches nothing under `variant` leaves them there
        outs.append(model.predict(x))
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    if precision == "f16x3":
        ref = kf.forward(g, w, x[:2])
        assert float(np.abs(outs[0][:2] - ref).max()) < TOL_SOFTMAX["f16x3"] and exact_label_check(ref, outs[0][:2])[1] == 0
    model.release()


# ------------------------------------------------------------------------------- seam 2 at full size
@pytest.mark.parametrize("classes,precision", [(2, "f16"), (4, "f16")])
def test_predict_448_matches_oracle(classes, precision):
    """The FAST fp16 mode (opt-in; the label-exact default is checked by test_predict_448_label_exact)."""
    cfg, w, g, model = make_model(classes, 448, 448, seed=classes, precision=precision, max_batch=4)
    x = (patches_from_page(448, 448, 2, seed=9) / 255.0).astype(np.float32)
    ref = kf.forward(g, w, x)
    got = model.predict(x)
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.allclose(got.sum(-1), 1.0, atol=1e-5)
    d, mism, bad = compare_probs(ref, got, TOL_SOFTMAX[precision])
    print(f"[448 C={classes} {precision}] max|dsoftmax|={d:.4f} label mismatches={mism}/{ref[...,0].size} outside tolerance band={bad}")
    assert d < TOL_SOFTMAX[precision] and bad == 0
    assert mism / ref[..., 0].size < TOL_LABEL_FRAC_F16
    model.release()


@pytest.mark.parametrize("classes,decisive,seed", [(2, False, 2), (4, False, 4), (2, True, 7)])
def test_predict_448_label_exact(classes, decisive, seed):
    """The label-exact mode (split fp16, SBBSEG_PREC_F16X3 -- the default of the Python seams) on the 448x448 nets:
    max|d softmax| <= 2e-3 against the fp32 oracle and labels identical wherever the oracle's top-2 margin
    exceeds 1e-3 (north_star: "argmax label map bit-exact", main.py:290)."""
    cfg, w, g, model = make_model(classes, 448, 448, seed=seed, precision="f16x3", max_batch=4, decisive=decisive)
    x = (patches_from_page(448, 448, 2, seed=9 if not decisive else 5) / 255.0).astype(np.float32)
    ref = kf.forward(g, w, x)
    got = model.predict(x)
    assert got.shape == ref.shape and got.dtype == np.float32
    d = float(np.abs(ref - got).max())
    mism, bad = exact_label_check(ref, got)
    print(f"[448 C={classes} decisive={decisive} f16x3] max|dsoftmax|={d:.2e} label mismatches={mism}/{ref[...,0].size} "
          f"with oracle margin > {EXACT_MARGIN}: {bad}")
    assert d <= TOL_SOFTMAX["f16x3"] and bad == 0
    model.release()


def test_predict_448_decisive_net():
    """Same check on a trained-like net (tools/synth_model._make_decisive: logit differences follow ink vs
    paper, bimodal) -- the regime real models are in: labels agree with the oracle on all but a few
    pixels, and every disagreement sits at a near-tie of the oracle."""
    cfg, w, g, model = make_model(2, 448, 448, seed=7, precision="f16", max_batch=4, decisive=True)
    x = (patches_from_page(448, 448, 2, seed=5) / 255.0).astype(np.float32)
    ref = kf.forward(g, w, x)
    got = model.predict(x)
    d, mism, bad = compare_probs(ref, got, TOL_SOFTMAX["f16"])
    print(f"[448 decisive f16] max|dsoftmax|={d:.4f} label mismatches={mism}/{ref[...,0].size} outside tolerance band={bad}")
    assert d < TOL_SOFTMAX["f16"] and bad == 0
    assert mism / ref[..., 0].size < 0.01
    model.release()


# ------------------------------------------------------------ seam 1: fused page path vs oracle loop
def test_segment_page_matches_oracle_loop():
    cfg, w, g, model = make_model(2, 224, 224, seed=5, precision="f16", max_batch=5)
    page = synthetic_page(500, 610, seed=3)
    om = kf.OracleModel(cfg, w)
    ref = tiling.do_prediction(True, page, om)                                   # oracle: main.py:225-366
    got = predict.do_prediction(True, page, model)
    assert got.dtype == np.uint8 and got.shape == ref.shape
    assert np.array_equal(got[:, :, 0], got[:, :, 1]) and np.array_equal(got[:, :, 0], got[:, :, 2])
    mism = (got[:, :, 0] != ref[:, :, 0]).mean()
    print(f"[page 500x610, 224 model] label mismatch fraction vs oracle loop: {mism:.5f}")
    assert mism < TOL_LABEL_FRAC_F16
    # the f32 check handle must agree with the oracle almost everywhere (only summation order differs)
    cfg2, w2, g2, m32 = make_model(2, 224, 224, seed=5, precision="f32", max_batch=5)
    got32 = predict.do_prediction(True, page, m32)
    assert (got32[:, :, 0] != ref[:, :, 0]).mean() < 5e-4
    # generic path (float page -> batched model.predict on the GPU -> host stitch) == fused path
    got_f = predict.do_prediction(True, page.astype(np.float64), model)
    assert np.array_equal(got_f, got)
    model.release(); m32.release()


def test_whole_image_branch_matches_oracle():
    cfg, w, g, model = make_model(2, 224, 224, seed=6, precision="f32", max_batch=2)
    page = synthetic_page(700, 520, seed=8)
    om = kf.OracleModel(cfg, w)
    ref = tiling.do_prediction(False, page, om, full_image_shape=(840, 624, 3))
    got = predict.do_prediction(False, page, model, full_image_shape=(840, 624, 3))
    assert got.shape == (840, 624, 3) and got.dtype == np.uint8
    # f32 tolerance: a label may differ only where the oracle's own top-2 softmax margin (at the model-resolution pixel the
    # output pixel is resized from) is inside 2 x the f32 softmax tolerance
    x = tiling.resize_nearest(page / 255.0, 224, 224)[None].astype(np.float32)
    pr = np.sort(om.predict(x)[0], axis=-1)
    margin = tiling.resize_nearest((pr[..., -1] - pr[..., -2])[:, :, None], 840, 624)[:, :, 0]
    mism = got[:, :, 0] != ref[:, :, 0]
    assert not (mism & (margin > 2 * TOL_SOFTMAX["f32"])).any()
    assert mism.mean() < 2e-4
    model.release()


# ------------------------------------------------- stitch / tile ranges vs the reference's own output
_SMALL_STITCH_MODELS = {}


def _stitch_handle(mh, mw, stitch_model):
    """A handle whose model input is mh x mw: stitch_kernel's owner tables come from the HANDLE's tile size and margin (the margin from
    the WIDTH for both axes, main.py:233) -- one small handle per model size of the fixture, kept for the module."""
    if (mh, mw) == (448, 448):
        return stitch_model
    if (mh, mw) not in _SMALL_STITCH_MODELS:
        _SMALL_STITCH_MODELS[(mh, mw)] = make_model(2, mh, mw, seed=1, precision="f16", max_batch=2, calib_hw=64)[3]
    return _SMALL_STITCH_MODELS[(mh, mw)]


@pytest.mark.parametrize("case", GOLD, ids=lambda c: f"{c['page_h']}x{c['page_w']}_m{c['model_h']}x{c['model_w']}")
def test_device_stitch_reproduces_reference_fixture(case, torch_cuda, stitch_model):
    """ALL twelve fixture cases through stitch_kernel, the non-square 320 x 480 model and the 224 x 224 model included (round 6)."""
    torch = torch_cuda
    ph, pw, mh, mw = case["page_h"], case["page_w"], case["model_h"], case["model_w"]
    stitch_model = _stitch_handle(mh, mw, stitch_model)
    page = tiling.coord_page(ph, pw).astype(np.int64)
    yy, xx = np.mgrid[0:mh, 0:mw]
    tiles = np.empty((case["n_calls"], mh, mw), np.uint8)
    for k, (x0, y0) in enumerate(case["calls_xy"]):                            # FakeModel's label rule
        p = page[y0:y0 + mh, x0:x0 + mw]
        tiles[k] = (k * 5 + yy * 3 + xx * 7 + p[:, :, 0] + 2 * p[:, :, 1]) % 16
    d_tiles = torch.from_numpy(tiles).cuda()
    d_out = torch.zeros((ph, pw), dtype=torch.uint8, device="cuda")
    stitch_model.ctx.stitch_dev(d_tiles.data_ptr(), ph, pw, d_out.data_ptr())
    stitch_model.ctx.synchronize()
    out = d_out.cpu().numpy()
    assert zlib.crc32(out.tobytes()) & 0xFFFFFFFF == case["out_crc32"]
    assert int(out.astype(np.int64).sum()) == case["out_sum"]


@pytest.fixture(scope="module")
def stitch_model():
    from sbb_textline_detection_amd.model import SegModel
    from tools.synth_model import calibrated_model
    cfg, w = calibrated_model(2, 448, 448, seed=0)      # BN statistics calibrated: a non-trivial label map
    m = SegModel(cfg, w, device=0, max_batch=8)          # default precision = the label-exact f16x3 mode
    m.test_cfg, m.test_weights = cfg, w
    yield m
    m.release()


# -------------------------------------------------------------- BASELINE config[1]: size-independent
def test_full_page_properties_3500x2500(torch_cuda, stitch_model):
    """One 3500x2500 page, 448 model (70 tiles).  The oracle needs ~2 min for this, so check
    properties instead: determinism, independence of the tile batch size, tile-range sharding ==
    whole page, and oracle agreement on a sample of tiles."""
    torch = torch_cuda
    model = stitch_model
    page = synthetic_page(3500, 2500, seed=0)
    a = model.segment_page(page)
    b = model.segment_page(page)
    assert a.shape == (3500, 2500) and np.array_equal(a, b)
    from sbb_textline_detection_amd.model import SegModel
    cfg, w = model.test_cfg, model.test_weights
    assert 0.02 < float(a.mean()) < 0.98, "calibrated net: both labels occur"
    m3 = SegModel(cfg, w, device=0, max_batch=3)
    assert np.array_equal(m3.segment_page(page), a)
    # sharded tile ranges, stitched, equal the one-call result
    d_page = torch.from_numpy(page).cuda()
    d_tiles = torch.empty((70, 448, 448), dtype=torch.uint8, device="cuda")
    for first, n in ((0, 23), (23, 23), (46, 24)):
        m3.ctx.segment_tile_range_dev(d_page.data_ptr(), 3500, 2500, first, n, d_tiles[first:].data_ptr())
    d_out = torch.empty((3500, 2500), dtype=torch.uint8, device="cuda")
    m3.ctx.stitch_dev(d_tiles.data_ptr(), 3500, 2500, d_out.data_ptr())
    m3.ctx.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), a)
    m3.release()
    # oracle agreement on a sample of the 70 tiles (the oracle needs ~1 s per tile on the GPU box's host cores):
    # the page map inside a tile's owned region == argmax of the oracle's softmax for that tile, label-exact
    # (stitch_model runs the seams' default mode, f16x3)
    tiles, nxf, nyf = tiling.tile_grid(3500, 2500, 448, 448)
    assert len(tiles) == 70
    from oracle.keras_config import read_model_config
    g = read_model_config(cfg)
    differ = 0
    owner = tiling.owner_map(3500, 2500, 448, 448)
    GROUP = 7                                                        # ALL 70 tiles (round 6), seven per oracle call: ~1 s of oracle per tile
    for k0 in range(0, len(tiles), GROUP):
        group = tiles[k0:k0 + GROUP]
        x = np.stack([page[t["y0"]:t["y0"] + 448, t["x0"]:t["x0"] + 448] for t in group]).astype(np.float32) / np.float32(255.0)
        refs = kf.forward(g, w, x)
        for q, t in enumerate(group):
            k = k0 + q
            sl = (slice(t["y0"] + t["ylo"], t["y0"] + t["yhi"]), slice(t["x0"] + t["xlo"], t["x0"] + t["xhi"]))
            got_tile = a[sl]
            own = owner[sl] == k
            r = refs[q, t["ylo"]:t["yhi"], t["xlo"]:t["xhi"]]
            srt = np.sort(r, axis=-1)
            decided = (srt[..., -1] - srt[..., -2]) > EXACT_MARGIN
            bad = (got_tile != r.argmax(-1)) & own & decided
            differ += int(((got_tile != r.argmax(-1)) & own).sum())
            assert not bad.any(), f"tile {k}: {int(bad.sum())} labels differ from the oracle outside its near-ties"
    print(f"[3500x2500, all 70 tiles vs the oracle] {differ} of {3500 * 2500} labels differ, all inside the oracle's near-ties")
    assert differ <= 1e-4 * 3500 * 2500


def test_sharded_entry_points_single_rank(torch_cuda, stitch_model):
    """distributed.segment_page_sharded / segment_pages_sharded on one rank == the one-call fused path."""
    from sbb_textline_detection_amd import distributed as D
    page = synthetic_page(1000, 900, seed=4)
    be = D.DeviceBackend(stitch_model)
    n_tiles = _capi.tile_grid(1000, 900, 448, 448)[0].shape[0]
    a = D.segment_page_sharded(be, be.to_device(page), n_tiles).cpu().numpy()
    b = D.segment_pages_sharded(be, [page, page[::-1].copy()]).cpu().numpy()
    ref = stitch_model.segment_page(page)
    assert np.array_equal(a, ref) and np.array_equal(b[0], ref)
    assert np.array_equal(b[1], stitch_model.segment_page(page[::-1].copy()))
    stitch_model.ctx.synchronize()
    stitch_model.ctx.set_stream(-1)


def test_scaled_page_equals_resized_page(stitch_model):
    """get_image_and_scales fused into the gather: segment_page_scaled(stored page) == segment_page(resized page)."""
    from sbb_textline_detection_amd.predict import resize_nearest
    from sbb_textline_detection_amd.stages import scaled_size
    page = synthetic_page(900, 700, seed=6)
    hs, ws = 1200, 933
    a = stitch_model.ctx.segment_page_scaled(page, hs, ws)
    b = stitch_model.segment_page(np.ascontiguousarray(resize_nearest(page, hs, ws)))
    assert a.shape == (hs, ws) and np.array_equal(a, b)
    assert scaled_size(900, 700) == (2800, 2177)


def test_three_model_pipeline(tmp_path):
    """BASELINE config[2]: border (whole image, 2 classes) + layout (Otsu'd page, 4 classes) + textline
    (2 classes) through the stage wrappers, models loaded from .sbbw files via the reference's .h5 paths."""
    from sbb_textline_detection_amd import clear_session, stages
    from sbb_textline_detection_amd.weights import save_sbbw
    from tools.synth_model import calibrated_model
    specs = {"model_page_mixed_best": 2, "model_strukturerkennung": 4, "model_textline_new": 2}      # main.py:58-60
    models = {}
    for name, classes in specs.items():
        cfg, w = calibrated_model(classes, 224, 224, seed=classes)
        save_sbbw(str(tmp_path / (name + ".sbbw")), cfg, w)
        models[name] = (cfg, w)
    st = stages.InferenceStages(*[str(tmp_path / (n + ".h5")) for n in specs], model_kwargs={"max_batch": 16})
    page = synthetic_page(520, 400, seed=9)                                   # < 2500 high -> upscaled to 2800 x 2153
    mask, regions, lines, page_coord = st.run(page)
    hs, ws = stages.scaled_size(520, 400)
    bx, by, bw, bh = st.page_box
    assert page_coord == [by, by + bh, bx, bx + bw] and bw >= 224 and bh >= 224
    assert mask.shape == (hs, ws, 3) and regions.shape == (bh, bw, 3) and lines.shape == (bh, bw)
    assert mask.dtype == regions.dtype == lines.dtype == np.uint8
    assert regions.max() <= 3 and lines.max() <= 1 and mask.max() <= 1
    # The chaining of run() (main.py:2061-2102): both patch stages see the CROPPED upscaled page.  The fused path (rescale +
    # crop + Otsu in the tile gather, nothing materialised) == the reference's sequence on materialised arrays:
    # resize -> crop_image_inside_box -> otsu_copy -> astype(uint8) -> do_prediction -> erode x 3 / dilate x 4
    from oracle import stage_glue
    from sbb_textline_detection_amd.model import load_model
    from sbb_textline_detection_amd.predict import resize_nearest
    up = resize_nearest(page, hs, ws)
    pbox, ppix = stage_glue.page_box(mask)
    assert tuple(pbox) == (bx, by, bw, bh)                                   # device box == oracle box of the same mask
    crop, coord = stage_glue.crop_image_inside_box((bx, by, bw, bh), up)
    assert coord == page_coord
    ots = stage_glue.otsu_copy(crop).astype(np.uint8)
    assert st.otsu_threshold == stage_glue.otsu_threshold(crop[:, :, 0])
    layout = load_model(str(tmp_path / "model_strukturerkennung.h5"), max_batch=16)          # default precision: label-exact f16x3
    regions2 = predict.do_prediction(True, ots, layout)
    assert np.array_equal(regions, stage_glue.region_cleanup(regions2))
    lines2 = predict.do_prediction(True, crop, load_model(str(tmp_path / "model_textline_new.h5"), max_batch=16))[:, :, 0]
    assert np.array_equal(lines, lines2)
    # the un-cropped forms still equal their materialised twins
    assert np.array_equal(st.textline_contours(None), predict.do_prediction(True, up, load_model(str(tmp_path / "model_textline_new.h5"), max_batch=16))[:, :, 0])
    # label-exact: probabilities of three tiles of the Otsu'd crop against the oracle, labels equal outside EXACT_MARGIN
    cfg, w = models["model_strukturerkennung"]
    om = kf.OracleModel(cfg, w)
    ys, xs_ = (0, (bh - 224) // 2, bh - 224), (0, (bw - 224) // 2, bw - 224)
    xs = np.stack([ots[y0:y0 + 224, x0:x0 + 224] for (y0, x0) in zip(ys, xs_)]).astype(np.float32) / np.float32(255.0)
    pref, pgot = om.predict(xs), layout.predict(xs)
    assert float(np.abs(pref - pgot).max()) < TOL_SOFTMAX["f16x3"]
    total, outside = exact_label_check(pref, pgot)
    assert outside == 0 and total <= 1e-3 * pref[..., 0].size
    clear_session()


def test_full_size_three_model_pipeline_config3(tmp_path):
    """BASELINE configs[2] at FULL size: border (448x448, 2 classes, whole image) + layout (448x448, 4 classes, Otsu'd crop) +
    textline (448x448, 2 classes) on one 3500x2500 page -> upscaled to 4200x3000 (main.py:205-207) -> up to 1 + 108 + 108
    forwards.  Label-exact default mode; the layout and textline maps are checked against the oracle on sampled tiles of the
    cropped page (owned pixels, outside EXACT_MARGIN), the glue against the oracle's restatements."""
    from oracle import stage_glue
    from sbb_textline_detection_amd import clear_session, stages
    from sbb_textline_detection_amd.predict import resize_nearest
    from sbb_textline_detection_amd.weights import save_sbbw
    from tools.synth_model import calibrated_model
    specs = {"model_page_mixed_best": (2, 21), "model_strukturerkennung": (4, 22), "model_textline_new": (2, 23)}      # main.py:58-60
    models = {}
    for name, (classes, seed) in specs.items():
        cfg, w = calibrated_model(classes, 448, 448, seed=seed)
        save_sbbw(str(tmp_path / (name + ".sbbw")), cfg, w)
        models[name] = (cfg, w)
    st = stages.InferenceStages(*[str(tmp_path / (n + ".h5")) for n in specs], model_kwargs={"max_batch": 108})
    page = synthetic_page(3500, 2500, seed=33)
    st.get_image_and_scales(page)
    assert (st.img_hight_int, st.img_width_int) == (4200, 3000)
    mask, box, page_coord = st.page_box_only()
    bx, by, bw, bh = box
    assert mask.shape == (4200, 3000, 3) and tuple(stage_glue.page_box(mask)[0]) == box and bw >= 448 and bh >= 448
    regions_raw = st.extract_text_regions(box=box)
    lines = st.textline_contours(box=box)
    assert regions_raw.shape == (bh, bw, 3) and lines.shape == (bh, bw) and regions_raw.max() <= 3 and lines.max() <= 1
    regions = st.clean_text_regions(regions_raw)
    assert np.array_equal(regions, stage_glue.region_cleanup(regions_raw))                       # main.py:2074-2075
    up = resize_nearest(page, 4200, 3000)
    crop = up[by:by + bh, bx:bx + bw]
    assert st.otsu_threshold == stage_glue.otsu_threshold(crop[:, :, 0])
    ots = stage_glue.otsu_copy(crop).astype(np.uint8)
    tiles, nxf, nyf = tiling.tile_grid(bh, bw, 448, 448)
    own = tiling.owner_map(bh, bw, 448, 448)
    print(f"[config 3, full size] box {box}: {len(tiles)} tiles per patch stage, {1 + 2 * len(tiles)} forwards")
    picks = sorted({0, len(tiles) // 2, len(tiles) - 1})
    for name, src_img, got in (("model_strukturerkennung", ots, regions_raw[:, :, 0]), ("model_textline_new", crop, lines)):
        cfg, w = models[name]
        for k in picks[:3] if name == "model_strukturerkennung" else picks[:2]:
            t = tiles[k]
            x = (src_img[t["y0"]:t["y0"] + 448, t["x0"]:t["x0"] + 448][None] / 255.0).astype(np.float32)
            ref = kf.forward_config(cfg, w, x)[0]
            sl = (slice(t["y0"] + t["ylo"], t["y0"] + t["yhi"]), slice(t["x0"] + t["xlo"], t["x0"] + t["xhi"]))
            r = ref[t["ylo"]:t["yhi"], t["xlo"]:t["xhi"]]
            srt = np.sort(r, axis=-1)
            mism = (got[sl] != r.argmax(-1)) & (own[sl] == k)
            bad = mism & ((srt[..., -1] - srt[..., -2]) > EXACT_MARGIN)
            assert not bad.any(), (name, k, int(bad.sum()))
            assert mism.mean() < 1e-3, (name, k, float(mism.mean()))
    clear_session()


def test_c_abi_error_paths_do_not_abort(stitch_model):
    """Bad arguments come back as return codes / RuntimeError (the reference's callers rely on
    ordinary exceptions, main.py:2061-2157) and leave the handle usable."""
    m = stitch_model
    with pytest.raises(RuntimeError, match="smaller than the model"):
        m.segment_page(np.zeros((300, 500, 3), np.uint8))
    with pytest.raises(ValueError):
        m.predict(np.zeros((1, 100, 100, 3), np.float32))
    with pytest.raises(RuntimeError):
        m.ctx.debug_read_tensor(0, 10 ** 6, (1, 1, 1))
    with pytest.raises(RuntimeError, match="variant"):
        m.ctx.set_conv_variant(1 << 26)
    page = synthetic_page(448, 448, seed=1)              # 448x448 page -> 4 identical clamped tiles
    lab = m.segment_page(page)
    assert lab.shape == (448, 448)
    one = m.predict((page[None] / 255.0).astype(np.float32)).argmax(-1)[0].astype(np.uint8)
    assert np.array_equal(lab, one)
    # round 5's entry points: bad arguments are status codes too, and the handle survives them
    import ctypes as C
    lib, h = m.ctx.lib, m.ctx.h
    info = _capi.RunInfo()
    buf = np.zeros(16, np.uint8)
    assert lib.sbbseg_run_page(h, h, h, None, 10, 10, 10, 10, 3, None, _capi._ptr(buf), _capi._ptr(buf), C.byref(info)) != 0      # no page
    assert lib.sbbseg_run_page(h, h, None, _capi._ptr(buf), 2, 2, 2, 2, 3, None, _capi._ptr(buf), _capi._ptr(buf), C.byref(info)) != 0   # null handle
    assert lib.sbbseg_run_page(h, h, h, _capi._ptr(buf), 2, 2, 2, 2, 2, None, _capi._ptr(buf), _capi._ptr(buf), C.byref(info)) != 0      # channels
    p_out = C.c_void_p()
    assert lib.sbbseg_device_alloc(h, 0, C.byref(p_out)) != 0 and lib.sbbseg_device_alloc(None, 16, C.byref(p_out)) != 0
    assert lib.sbbseg_device_free(h, C.c_void_p(0x1000)) != 0 and b"not allocated" in lib.sbbseg_last_error()
    pres = C.c_int(7)
    assert lib.sbbseg_text_regions_present_dev(h, None, 4, 4, 1, C.c_double(1e-5), C.byref(pres)) != 0
    # a page smaller than the layout model: the border stage works on any size, the layout stage fails like main.py:278-285 and the call
    # still succeeds with "no regions" (main.py:2089-2091)
    tiny = synthetic_page(300, 200, seed=3)
    mask, regions, lines, info = _capi.run_page(m.ctx, m.ctx, m.ctx, tiny, 300, 200, channels=1)
    assert mask.shape == (300, 200) and (info.regions_ok, info.text_present, info.textlines_ok) == (0, 0, 0) and regions is None and lines is None
    assert np.array_equal(m.segment_page(page), lab)


def test_batch_one_and_four_classes():
    cfg, w, g, model = make_model(4, 224, 224, seed=8, precision="f16", max_batch=1)
    page = synthetic_page(400, 460, seed=2)
    a = predict.do_prediction(True, page, model)
    cfg2, w2, g2, m7 = make_model(4, 224, 224, seed=8, precision="f16", max_batch=7)
    b = predict.do_prediction(True, page, m7)
    assert np.array_equal(a, b) and a.max() <= 3
    model.release(); m7.release()


def test_device_otsu_threshold_and_binarised_gather(stitch_model):
    """otsu_copy on the device: histogram + getThreshVal_Otsu_8u arithmetic == the oracle's
    threshold bit for bit, and the binarising tile gather == segmenting the host-binarised page."""
    from oracle import stage_glue
    from sbb_textline_detection_amd.predict import resize_nearest
    m = stitch_model
    rng = np.random.RandomState(3)
    cases = [synthetic_page(600, 520, seed=1), synthetic_page(901, 777, seed=2),
             rng.randint(0, 256, (500, 470, 3)).astype(np.uint8),                       # flat histogram
             np.full((460, 450, 3), 200, np.uint8)]                                     # constant page -> threshold 0
    cases[1][:, :, 1] = 255 - cases[1][:, :, 1]                                          # only channel 0 may matter
    for page in cases:
        lab, thr = m.ctx.segment_page_otsu(page)
        assert thr == stage_glue.otsu_threshold(page[:, :, 0])
        ref = m.segment_page(stage_glue.otsu_copy(page).astype(np.uint8))
        assert np.array_equal(lab, ref)
    page = cases[0]
    hs, ws = 812, 703                                                                    # through the nearest rescale
    lab, thr = m.ctx.segment_page_otsu(page, hs, ws)
    up = resize_nearest(page, hs, ws)
    assert thr == stage_glue.otsu_threshold(up[:, :, 0])
    assert np.array_equal(lab, m.segment_page(stage_glue.otsu_copy(up).astype(np.uint8)))


def test_device_otsu_building_blocks(stitch_model):
    """sbbseg_otsu_dev + sbbseg_segment_tile_range_bin_dev (the sharded form) == the one-call form."""
    import torch
    from oracle import stage_glue
    m = stitch_model
    page = synthetic_page(700, 640, seed=5)
    Hp, Wp = page.shape[:2]
    H, W, _, _ = m.ctx.model_info()
    d_page = torch.from_numpy(page).cuda()
    d_thr = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        m.ctx.otsu_dev(d_page.data_ptr(), Hp, Wp, d_thr.data_ptr())
        from sbb_textline_detection_amd._capi import tile_grid
        _, nx, ny = tile_grid(Hp, Wp, H, W)
        n = nx * ny
        d_tiles = torch.empty((n, H, W), dtype=torch.uint8, device="cuda")
        half = n // 2
        m.ctx.segment_tile_range_bin_dev(d_page.data_ptr(), Hp, Wp, 0, half, d_thr.data_ptr(), d_tiles.data_ptr())
        m.ctx.segment_tile_range_bin_dev(d_page.data_ptr(), Hp, Wp, half, n - half, d_thr.data_ptr(), d_tiles[half:].data_ptr())
        d_lab = torch.empty((Hp, Wp), dtype=torch.uint8, device="cuda")
        m.ctx.stitch_dev(d_tiles.data_ptr(), Hp, Wp, d_lab.data_ptr())
        torch.cuda.synchronize()
    finally:
        m.ctx.set_stream(-1)
    assert int(d_thr.item()) == stage_glue.otsu_threshold(page[:, :, 0])
    lab, _ = m.ctx.segment_page_otsu(page)
    assert np.array_equal(d_lab.cpu().numpy(), lab)


def test_two_lanes_equal_one_lane():
    """The two-lane split of a chunk (second half on a private stream with its own buffers) changes
    nothing but the schedule: identical label maps, for plain / rescaled / Otsu page entry points."""
    cfg, w, g, model = make_model(2, 224, 224, seed=3, precision="f16", max_batch=40)
    page = synthetic_page(1400, 1100, seed=7)                      # 8 x 7 = 56 tiles at 224: chunks of 40 + 16
    outs = {}
    for lanes in (2, 1, 2):
        model.ctx.set_lanes(lanes)
        outs[lanes] = (model.segment_page(page), model.ctx.segment_page_scaled(page, 1500, 1201),
                       model.ctx.segment_page_otsu(page)[0])
        if lanes == 1:
            ref = outs[1]
    for a, b in zip(outs[2], ref):
        assert np.array_equal(a, b)
    # the same on torch's current (legacy default) stream with device-resident buffers: how bench.py and the
    # sharded backend drive the library; the second lane forks from / joins that stream with events
    import torch
    d_page = torch.from_numpy(page).cuda()
    d_lab = torch.zeros(page.shape[:2], dtype=torch.uint8, device="cuda")
    model.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for _ in range(3):
            model.ctx.segment_page_dev(d_page.data_ptr(), page.shape[0], page.shape[1], d_lab.data_ptr())
        got = d_lab.cpu().numpy()                      # (.cpu() synchronises the current stream only)
    finally:
        model.ctx.set_stream(-1)
    assert np.array_equal(got, ref[0])
    model.release()


def test_three_channel_host_output(stitch_model):
    """label_channels = 3: the reference's return layout (uint8 [H,W,3], equal channels) comes off the device."""
    m = stitch_model
    for shape in ((500, 610), (449, 451)):                      # odd pixel counts: the replicate kernel works on 4-label words
        page = synthetic_page(shape[0], shape[1], seed=3)
        one = m.segment_page(page)
        three = m.segment_page(page, channels=3)
        assert three.shape == shape + (3,) and three.dtype == np.uint8
        for ch in range(3):
            assert np.array_equal(three[:, :, ch], one)
        w1 = m.segment_whole(page, 333, 257)
        w3 = m.segment_whole(page, 333, 257, channels=3)
        assert np.array_equal(w3, np.repeat(w1[:, :, None], 3, axis=2))
        assert np.array_equal(m.segment_page(page), one)        # and back to one plane
    out = predict.do_prediction(True, page, m)
    assert out.shape == page.shape and np.array_equal(out[:, :, 1], one)


def test_whole_image_branch_through_composed_rescale(stitch_model):
    """sbbseg_segment_whole_scaled(stored page) == sbbseg_segment_whole(nearest-upscaled page): the border stage
    (main.py:384-392) without building the upscaled page."""
    from sbb_textline_detection_amd.predict import resize_nearest
    m = stitch_model
    page = synthetic_page(611, 503, seed=8)
    hs, ws = 1234, 1017
    a = m.ctx.segment_whole_scaled(page, hs, ws, hs, ws)
    b = m.segment_whole(np.ascontiguousarray(resize_nearest(page, hs, ws)), hs, ws)
    assert a.shape == (hs, ws) and np.array_equal(a, b)


@pytest.mark.parametrize("precision,hw,max_batch", [("f16", (1400, 1200), 24), ("f16x3", (3500, 2500), 70), ("f16", (3500, 2500), 70)])
def test_repeatability_under_load(precision, hw, max_batch):
    """Race screen for the hand-placed waits (counted vmcnt with stores in flight, LDS-DMA staging, phase-shifted wave groups of the
    split tail, mid-K-step load issue, two lanes): the same page segmented 12 times back to back must give bit-identical maps (a
    stage read before it landed shows up as run-to-run differences long before it shows up as a parity failure).  The f16x3 case
    runs the mode and the launch shapes the bench number is quoted on: a 3500x2500 page = 70 tiles in ONE chunk = two lanes of 35
    tiles (>= 32 per lane: every persistent grid is full, 512x128 / 256x256 tiles, block_x3, dec_tail_fused_x3ps all engaged)."""
    cfg, w, g, model = make_model(2, 448, 448, seed=0, precision=precision, max_batch=max_batch)
    page = synthetic_page(hw[0], hw[1], seed=11)
    first = model.segment_page(page)
    assert 0.02 < float(first.mean()) < 0.98
    for _ in range(11):
        assert np.array_equal(model.segment_page(page), first)
    model.ctx.set_lanes(1)
    assert np.array_equal(model.segment_page(page), first)
    model.release()


def test_random_page_sizes_fused_equals_reference_loop():
    """Randomised geometry sweep: for page sizes the fixtures do not cover, the fused device path (tile gather,
    batched forward, argmax, stitch kernel, two lanes) must equal the oracle's restatement of the reference
    loop driven with the SAME model through seam 2 (model.predict per tile + host argmax + host paste) --
    bit for bit, because both sides run the same kernels on the same tiles."""
    cfg, w, g, model = make_model(4, 224, 224, seed=9, precision="f16", max_batch=20)
    rng = np.random.RandomState(1234)
    sizes = [(224, 224), (224 + 1, 224 * 3), (int(rng.randint(230, 900)), int(rng.randint(230, 900))),
             (int(rng.randint(230, 900)), int(rng.randint(230, 900))), (180 * 3, 180 * 4), (180 * 3 + 1, 180 * 4 - 1)]
    for hp, wp in sizes:
        page = synthetic_page(hp, wp, seed=hp * 7 + wp)
        fused = predict.do_prediction(True, page, model)
        loop = tiling.do_prediction(True, page, model)            # oracle tiling code, HIP model behind model.predict
        assert fused.shape == (hp, wp, 3) and np.array_equal(fused, loop), (hp, wp)
    model.release()


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_committed_forward_fixture(precision):
    """tests/golden/forward_golden_64.npz (seeded 64x64 net; inputs + fp32 probabilities committed with its
    generator) through the C ABI: fp32 check mode within TOL_SOFTMAX['f32'], fp16 within its band."""
    import os
    from sbb_textline_detection_amd.model import SegModel
    from sbb_textline_detection_amd.weights import synthetic_model
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "forward_golden_64.npz"))
    cfg, w = synthetic_model(int(d["classes"]), 64, 64, seed=int(d["seed"]))
    m = SegModel(cfg, w, device=0, max_batch=2, precision=precision)
    p = m.predict(d["x"])
    assert p.shape == d["probs"].shape and p.dtype == np.float32
    err = float(np.abs(p - d["probs"]).max())
    print(f"[golden 64x64, {precision}] max|dsoftmax| = {err:.2e}")
    assert err < TOL_SOFTMAX[precision]
    if precision == "f32":
        srt = np.sort(d["probs"], axis=-1)
        decided = (srt[..., -1] - srt[..., -2]) > 1e-3
        assert np.array_equal(p.argmax(-1)[decided], d["probs"].argmax(-1)[decided])
    m.release()


def test_config4_pages_4000x3000_sharded_single_rank(torch_cuda, stitch_model):
    """BASELINE config[3] at one GPU: pages of 4000x3000 (108 tiles each) through distributed.segment_pages_sharded ==
    the one-call fused path per page, and the label-exact mode agrees with the oracle on sampled tiles."""
    from sbb_textline_detection_amd import distributed as D
    pages = [synthetic_page(4000, 3000, seed=100), synthetic_page(4000, 3000, seed=101)]
    be = D.DeviceBackend(stitch_model)
    got = D.segment_pages_sharded(be, pages).cpu().numpy()
    stitch_model.ctx.synchronize()
    stitch_model.ctx.set_stream(-1)
    assert got.shape == (2, 4000, 3000)
    w = stitch_model.test_weights
    tiles, nxf, nyf = tiling.tile_grid(4000, 3000, 448, 448)
    assert len(tiles) == 108
    own = tiling.owner_map(4000, 3000, 448, 448)
    for p, page in enumerate(pages):
        assert np.array_equal(got[p], stitch_model.segment_page(page))
        for k in (5 + 40 * p, 107 - 30 * p):
            t = tiles[k]
            x = (page[t["y0"]:t["y0"] + 448, t["x0"]:t["x0"] + 448][None] / 255.0).astype(np.float32)
            ref = kf.forward_config(stitch_model.test_cfg, w, x)[0]
            sl = (slice(t["y0"] + t["ylo"], t["y0"] + t["yhi"]), slice(t["x0"] + t["xlo"], t["x0"] + t["xhi"]))
            r = ref[t["ylo"]:t["yhi"], t["xlo"]:t["xhi"]]
            srt = np.sort(r, axis=-1)
            bad = (got[p][sl] != r.argmax(-1)) & (own[sl] == k) & ((srt[..., -1] - srt[..., -2]) > EXACT_MARGIN)
            assert not bad.any(), (p, k, int(bad.sum()))


def test_scaled_page_equals_oracle_resize(stitch_model):
    """SURVEY 8f-2 (get_image_and_scales fused into the gather): segment_page_scaled(stored page) == segment_page of the
    page resized by the ORACLE's resize_nearest (not the product's), at sizes the reference's own rules produce
    (main.py:201-207: the < 2500 -> 2800 rule and the x1.2 rule) and at an odd ratio."""
    from sbb_textline_detection_amd.stages import scaled_size
    m = stitch_model
    for (h, w, hs, ws) in ((700, 560, 2800, 2240), (2600, 500, 3120, 600), (611, 503, 1234, 1017)):
        if (hs, ws) != (1234, 1017):
            assert scaled_size(h, w) == (hs, ws)
        page = synthetic_page(h, w, seed=h)
        a = m.ctx.segment_page_scaled(page, hs, ws)
        b = m.segment_page(np.ascontiguousarray(tiling.resize_nearest(page, hs, ws)))
        assert a.shape == (hs, ws) and np.array_equal(a, b), (h, w, hs, ws)


def test_whole_image_branch_fused_equals_reference_structure(stitch_model):
    """patches=False (main.py:368-380): the fused device path (resize gather -> forward -> argmax -> resize back) must equal,
    bit for bit, the oracle's restatement of the branch -- itself pinned to the imported reference by the whole_cases
    fixtures -- driven with the SAME HIP model through seam 2 (model.predict): same net, same kernels, so any
    difference is a structural one (which size is resized to which, main.py:378's self.image.shape)."""
    m = stitch_model
    m.ctx.set_ksplit(False)      # "same kernels": the branch's split-K launches (round 4) differ from seam 2's in the last bits; they have
    try:                         # their own test (test_whole_image_branch_split_k_matches_oracle_and_the_unsplit_launches)
        for (h, w, fh, fw) in ((700, 520, 840, 624), (611, 503, 2800, 2305), (448, 448, 448, 448), (1234, 777, 333, 257)):
            page = synthetic_page(h, w, seed=h + w)
            fused = predict.do_prediction(False, page, m, full_image_shape=(fh, fw, 3))
            loop = tiling.do_prediction(False, page, m, full_image_shape=(fh, fw, 3))
            assert fused.shape == (fh, fw, 3) and fused.dtype == np.uint8 and np.array_equal(fused, loop), (h, w, fh, fw)
    finally:
        m.ctx.set_ksplit(True)


def test_model_load_survives_injected_bad_alloc():
    """A std::bad_alloc inside the plan upload (sbbseg_add_conv) is a RuntimeError, not a dead process; the next load works."""
    lib = _capi.load_library()
    cfg, w, g, model = make_model(2, 64, 64, seed=1, precision="f16", max_batch=2, calib_hw=64)
    model.release()
    from sbb_textline_detection_amd.model import SegModel
    assert lib.sbbseg_debug_inject_alloc_failure(7) == 0
    with pytest.raises(RuntimeError, match="out of host memory"):
        SegModel(cfg, w, device=0, max_batch=2, precision="f16")
    assert lib.sbbseg_debug_inject_alloc_failure(0) == 0
    m2 = SegModel(cfg, w, device=0, max_batch=2, precision="f16")
    assert m2.predict(np.zeros((1, 64, 64, 3), np.float32)).shape == (1, 64, 64, 2)
    m2.release()


@pytest.mark.parametrize("precision", ["f16x3", "f32", "f16"])
def test_handwritten_keras23_fixture_through_c_abi(precision):
    """tests/golden/keras23_model_config.json (hand-written Keras-2.3 functional Model: marshalled Lambda, three ZeroPadding2D
    tuple forms, stride-2 1x1 projection, 3 classes) loaded by the product's parser/planner and run through the C ABI, against
    the oracle reading the same JSON with its OWN reader (oracle/keras_config.py)."""
    from sbb_textline_detection_amd.keras_graph import parse_model_config
    from sbb_textline_detection_amd.model import SegModel
    from sbb_textline_detection_amd.weights import synthetic_weights
    cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "keras23_model_config.json")))
    w = synthetic_weights(parse_model_config(cfg), seed=5)
    rng = np.random.RandomState(2)
    for name in list(w):
        if name.endswith("moving_variance:0"):
            w[name] = rng.uniform(0.5, 2.0, w[name].shape).astype(np.float32)
    x = rng.rand(3, 32, 48, 3).astype(np.float32)
    ref = kf.forward_config(cfg, w, x)
    m = SegModel(cfg, w, device=0, max_batch=3, precision=precision)
    got = m.predict(x)
    d = float(np.abs(got - ref).max())
    print(f"[keras23 fixture {precision}] max|dsoftmax| = {d:.2e}")
    assert got.shape == (3, 32, 48, 3)
    if precision in ("f16x3", "f32"):
        mism, bad = exact_label_check(ref, got)
        assert d < TOL_SOFTMAX[precision] and bad == 0
    else:
        assert d < 0.05
    m.release()


@pytest.mark.parametrize("k,precision", [(2, "f16x3"), (3, "f16x3"), (2, "f16"), (3, "f32")])
def test_conv2d_transpose_decoder_through_c_abi(k, precision):
    """A U-Net whose decoder upsamples with Conv2DTranspose (k x k, stride 2) -- lowered to output-parity class convs --
    through the C ABI against the oracle's scatter-form Conv2DTranspose."""
    from sbb_textline_detection_amd.keras_graph import parse_model_config, transpose_unet_config
    from sbb_textline_detection_amd.model import SegModel
    from sbb_textline_detection_amd.weights import synthetic_weights
    cfg = transpose_unet_config(3, 64, 96, k=k)
    w = synthetic_weights(parse_model_config(cfg), seed=4)
    x = (patches_from_page(64, 96, 3, seed=6) / 255.0).astype(np.float32)
    ref = kf.forward_config(cfg, w, x)
    m = SegModel(cfg, w, device=0, max_batch=3, precision=precision)
    got = m.predict(x)
    d = float(np.abs(got - ref).max())
    print(f"[convT k={k} {precision}] max|dsoftmax| = {d:.2e}")
    if precision in ("f16x3", "f32"):
        assert d < TOL_SOFTMAX[precision] and exact_label_check(ref, got)[1] == 0
    else:
        assert d < 0.05
    m.release()


def test_pooled_pages_equal_page_by_page(torch_cuda):
    """sbbseg_segment_pages_dev (tiles of several pages pooled into max_batch-sized chunks, chunks spanning page borders,
    two lanes) == sbbseg_segment_page_dev page by page, bit for bit."""
    torch = torch_cuda
    cfg, w, g, model = make_model(2, 224, 224, seed=3, precision="f16", max_batch=40)
    pages = [synthetic_page(700, 610, seed=20 + k) for k in range(5)]            # 4 x 4 = 16 tiles each -> chunks of 40 span pages
    d_pages = [torch.from_numpy(p).cuda() for p in pages]
    d_out = [torch.zeros((700, 610), dtype=torch.uint8, device="cuda") for _ in pages]
    model.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        model.ctx.segment_pages_dev([p.data_ptr() for p in d_pages], 700, 610, [o.data_ptr() for o in d_out])
        got = [o.cpu().numpy() for o in d_out]
    finally:
        model.ctx.set_stream(-1)
    for p, a in zip(pages, got):
        assert np.array_equal(a, model.segment_page(p))
    with pytest.raises(ValueError):
        model.ctx.segment_pages_dev([], 700, 610, [])
    model.release()


def test_device_morphology_and_page_box(torch_cuda, stitch_model):
    """SURVEY 8f-3 remainder on the device, bit-exact against oracle/stage_glue.py (which applies the 5x5 kernel literally,
    iteration by iteration): erode x 3 / dilate x 4 of a layout map (main.py:2074-2075), dilate x 6 + largest component +
    bounding box of a border mask (main.py:394-404), incl. speckle noise, corner-touching blobs, page edges and the empty mask."""
    from oracle import stage_glue
    torch = torch_cuda
    m = stitch_model
    rng = np.random.RandomState(11)
    # layout-like map: classes 0..3 in blobs + noise
    lay = np.zeros((613, 877), np.uint8)
    lay[50:300, 60:500] = 1; lay[320:600, 100:800] = 2; lay[0:40, 700:877] = 3
    lay[rng.rand(*lay.shape) < 0.01] = rng.randint(0, 4)
    for op, name, it in ((0, "erode", 3), (1, "dilate", 4), (1, "dilate", 6), (0, "erode", 1)):
        assert np.array_equal(m.ctx.morph(lay, op, 5, it), stage_glue.morph(lay, name, 5, it)), (name, it)
    assert np.array_equal(m.ctx.morph(m.ctx.morph(lay, 0, 5, 3), 1, 5, 4), stage_glue.region_cleanup(lay))
    # widths that are multiples of 4 take the four-pixels-per-thread passes (round 4): the same map cut to 876 columns, and planes
    # narrower than the filter (the window is clipped on both sides at once)
    for plane in (np.ascontiguousarray(lay[:, :876]), np.ascontiguousarray(lay[:37, :12]), rng.randint(0, 4, (5, 8)).astype(np.uint8),
                  rng.randint(0, 256, (3, 4)).astype(np.uint8), rng.randint(0, 256, (90, 64)).astype(np.uint8)):
        for op, name, it in ((0, "erode", 3), (1, "dilate", 4), (1, "dilate", 6), (0, "erode", 1), (1, "dilate", 1)):
            assert np.array_equal(m.ctx.morph(plane, op, 5, it), stage_glue.morph(plane, name, 5, it)), (plane.shape, name, it)
    masks = []
    a = np.zeros((700, 520), np.uint8); a[40:660, 30:500] = 1; a[rng.rand(*a.shape) < 0.002] = 1; masks.append(a)     # page + specks
    b = np.zeros((300, 300), np.uint8); b[0:100, 0:100] = 1; b[125:200, 125:290] = 1; masks.append(b)                 # two blobs, 25 px apart -> merge after dilation
    c = np.zeros((300, 300), np.uint8); c[0:100, 0:100] = 1; c[126:200, 126:290] = 1; masks.append(c)                 # 26 px apart -> stay separate
    d = (rng.rand(257, 391) < 0.0005).astype(np.uint8); masks.append(d)                                               # sparse specks: many small components
    masks.append(np.zeros((64, 64), np.uint8))                                                                         # empty
    masks.append(np.ones((50, 70), np.uint8))                                                                          # full
    for k, mask in enumerate(masks):
        d_mask = torch.from_numpy(mask).cuda()
        got = m.ctx.page_box_dev(d_mask.data_ptr(), mask.shape[0], mask.shape[1])
        assert got == stage_glue.page_box(mask), (k, got, stage_glue.page_box(mask))


def test_page_box_ranking_is_repeatable_on_equal_blobs(stitch_model):
    """Equal-area blobs: the LAST in raster order wins (oracle/stage_glue.largest_component_box: np.argmax over OpenCV's reversed
    contour list) -- every time.  Round 5 found the device ranking picking one of three equal blobs at random on a 4200 x 3000 mask:
    the flatten pass of the component labelling raced with other threads' path halving and left a few pixels pointing at a non-root
    ancestor, so a blob's area came out short in some runs (kernels.hip cc_flatten_kernel).  Page-sized masks, many repeats; plus
    a mask whose largest blob beats the others by ONE cell, where a short count flips the ranking outright."""
    from oracle import stage_glue
    c = stitch_model.ctx
    rng = np.random.RandomState(5)
    H, W = 4200, 3000
    a = np.zeros((H, W), np.uint8)
    for (y, x) in ((795, 1743), (3167, 1301), (3692, 2560), (2000, 200)):
        a[y + 12:y + 21, x + 12:x + 19] = 1                                  # 9 x 7 seeds -> 33 x 31 blobs after dilate x 6
    b = a.copy()
    b[500:1700, 300:2700] = 1                                                # one big component beside the equal ones
    b[rng.rand(H, W) < 0.00002] = 1
    e = np.zeros((H, W), np.uint8)
    e[100:2000, 100:2901] = 1                                                # 1900 x 2801 ...
    e[2100:4000, 100:2900] = 1                                               # ... against 1900 x 2800: one column of cells less
    for k, mask in enumerate((a, b, e)):
        want = stage_glue.page_box(mask)
        d = c.device_alloc(mask.size)
        try:
            c.upload(d, mask)
            for it in range(12):
                got = c.page_box_dev(d, H, W)
                assert got == want, (k, it, got, want)
        finally:
            c.device_free(d)
    assert stage_glue.page_box(a)[0] == (2560, 3692, 31, 33)


def test_extract_page_stage(tmp_path):
    """extract_page (main.py:384-437) through the stage wrapper: border model on the upscaled page, box and crop; the box equals
    the oracle's box of the mask the same call returns."""
    from oracle import stage_glue
    from sbb_textline_detection_amd import clear_session, stages
    from sbb_textline_detection_amd.weights import save_sbbw
    from tools.synth_model import calibrated_model
    cfg, w = calibrated_model(2, 224, 224, seed=2)
    for name in ("model_page_mixed_best", "model_strukturerkennung", "model_textline_new"):
        save_sbbw(str(tmp_path / (name + ".sbbw")), cfg, w)
    st = stages.InferenceStages(*[str(tmp_path / (n + ".h5")) for n in ("model_page_mixed_best", "model_strukturerkennung", "model_textline_new")],
                                model_kwargs={"max_batch": 8})
    page = synthetic_page(520, 400, seed=9)
    st.get_image_and_scales(page)
    try:
        croped, coord = st.extract_page()
    except ValueError:
        croped, coord = None, None                     # an all-background border mask raises like the reference (main.py:401)
    if croped is not None:
        box, px = stage_glue.page_box(st.page_mask)
        assert coord == [box[1], box[1] + box[3], box[0], box[0] + box[2]] and croped.shape[:2] == (box[3], box[2])
        assert st.cont_page[0].shape == (4, 2)
    regions = st.extract_text_regions()
    assert np.array_equal(st.clean_text_regions(regions), stage_glue.region_cleanup(regions))
    clear_session()


@pytest.mark.parametrize("precision", ["f16", "f16x3", "f32"])
def test_one_call_native_load_equals_python_planned_context(tmp_path, precision):
    """sbbseg_model_load_file (the library's own graph reader + planner, csrc/loader.cpp) against a context the Python planner
    built through the step-by-step plan API: same op list, bit-identical probabilities and label maps (main.py:216-223)."""
    from sbb_textline_detection_amd.model import SegModel, load_model
    from sbb_textline_detection_amd.weights import save_sbbw
    from tools.synth_model import calibrated_model
    cfg, w = calibrated_model(2, 224, 224, seed=3)
    path = str(tmp_path / "model_textline_new.sbbw")
    save_sbbw(path, cfg, w)
    py = SegModel(cfg, w, device=0, max_batch=20, precision=precision)
    nat = SegModel(cfg, None, device=0, max_batch=20, precision=precision, sbbw_path=path)
    ops_py, ops_nat = py.ctx.ops(), nat.ctx.ops()
    assert [(o["name"], o["flops"], o["issued_flops"]) for o in ops_py] == [(o["name"], o["flops"], o["issued_flops"]) for o in ops_nat]
    assert py.ctx.device_bytes() == nat.ctx.device_bytes()
    x = (patches_from_page(224, 224, 3, seed=2) / 255.0).astype(np.float32)
    assert np.array_equal(py.predict(x), nat.predict(x))
    page = synthetic_page(700, 610, seed=5)
    assert np.array_equal(py.segment_page(page), nat.segment_page(page))
    assert nat.plan.macs_per_patch() == py.plan.macs_per_patch()              # (lazy Python plan of a natively loaded model)
    # the reference-shaped seam: start_new_session_and_model(<dir>/model.h5) resolves to the .sbbw and loads it with one C call
    m = load_model(str(tmp_path / "model_textline_new.h5"), max_batch=20, precision=precision)
    assert getattr(m, "_sbbw_path", None) == path and np.array_equal(m.predict(x), py.predict(x))
    lib = _capi.load_library()
    h = C_void = None
    import ctypes as C
    h = C.c_void_p()
    assert lib.sbbseg_model_load_file(b"/nonexistent/model.sbbw", 0, _capi.PRECISIONS[precision], 4, 0, C.byref(h)) != 0
    assert b"cannot open" in lib.sbbseg_last_error()
    py.release(); nat.release()
    from sbb_textline_detection_amd import clear_session
    clear_session()


# ----------------------------------------------------------------------------- stage glue: deskew search (f-4)
def _region_mask(h, w, seed):
    """A text-region-like mask: slanted bars of uneven length + a few specks."""
    rng = np.random.RandomState(seed)
    m = np.zeros((h, w), np.uint8)
    period = max(8, h // 9)
    for y in range(period // 2, h - period // 2, period):
        x0, x1 = rng.randint(0, w // 5), w - rng.randint(0, w // 5)
        for x in range(x0, x1):
            yy = y + int(0.06 * (x - w / 2))
            if 0 <= yy < h - period // 3:
                m[yy:yy + period // 3, x] = 1
    ys, xs = rng.randint(0, h, 12), rng.randint(0, w, 12)
    m[ys, xs] = 1
    return m


@pytest.mark.parametrize("h,w,seed", [(37, 53, 0), (120, 80, 1), (90, 211, 2), (1, 1, 3), (64, 64, 4)])
def test_deskew_profiles_equal_oracle(h, w, seed, stitch_model):
    """sbbseg_deskew_profiles (one launch for the whole sweep) vs oracle/deskew.py::row_profiles, bit for bit, on both angle
    sweeps of return_deskew_slope (main.py:1622, 1670) -- with the library's own rotation matrices handed to the oracle (numpy's
    and libm's cos/sin may differ in the last bit; test_oracle_deskew pins them to 1e-12 of each other)."""
    from oracle import deskew as dk
    from sbb_textline_detection_amd import _capi
    model = stitch_model
    m = _region_mask(h, w, seed) if h > 1 else np.ones((1, 1), np.uint8)
    side = _capi.deskew_side(h, w)
    angles = np.concatenate([np.linspace(-25, 25, 80), np.linspace(-90, -50, 30)])
    if h * w > 12000:
        angles = angles[::5]
    got = model.ctx.deskew_profiles(m, angles)
    assert got.shape == (len(angles), side) and got.dtype == np.int32
    sq = dk.padded_square(m)
    ref = np.stack([(dk.warp_affine_cubic_replicate(sq, _capi.rotation_matrix(side // 2, side // 2, a)) != 0).sum(axis=1) for a in angles])
    assert np.array_equal(got, ref), (np.abs(got - ref).max(), np.argwhere(got != ref)[:5])
    # explicit matrices = the same answer; an empty mask projects to nothing
    mats = np.stack([_capi.rotation_matrix(side // 2, side // 2, a) for a in angles[:7]])
    assert np.array_equal(model.ctx.deskew_profiles(m, matrices=mats), got[:7])
    assert not model.ctx.deskew_profiles(np.zeros((h, w), np.uint8), angles[:3]).any()


def test_return_deskew_slope_equals_oracle(stitch_model):
    """The whole search (stages.return_deskew_slope: device profiles + host peak logic) against the oracle's, on region masks
    skewed by known angles -- including one steep enough for the second sweep (main.py:1669-1716)."""
    from oracle import deskew as dk
    from sbb_textline_detection_amd import stages
    model = stitch_model
    base = np.zeros((100, 170), np.uint8)
    for y in range(12, 88, 15):
        base[y:y + 6, 10:160] = 1
    sq = dk.padded_square(base)
    for true, sigma in ((4.0, 1.0), (-9.5, 1.0), (0.0, 2.0), (-70.0, 1.0)):
        skewed = (dk.rotate_image(sq, true) != 0).astype(np.uint8)
        got = stages.return_deskew_slope(skewed, sigma, ctx=model.ctx)
        ref = dk.return_deskew_slope(skewed, sigma)
        print(f"[deskew] skew {true:+.1f}: device {got:+.3f}, oracle {ref:+.3f}")
        assert got == ref
        if abs(true) < 15:
            assert abs(got + true) < 1.0


def test_deskew_golden_vectors(stitch_model):
    """The device path against the COMMITTED vectors (tests/golden/deskew_golden.npz), not a freshly run oracle."""
    from sbb_textline_detection_amd import stages
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "deskew_golden.npz"))
    for k in range(3):
        m = g[f"mask{k}"]
        assert np.array_equal(stitch_model.ctx.deskew_profiles(m, g["angles"]), g[f"counts{k}"])
        assert stages.return_deskew_slope(m, 1.0, ctx=stitch_model.ctx) == float(g[f"slope{k}"])


# ----------------------------------------------------------------------------- pipelined multi-page host path
@pytest.mark.parametrize("channels", [1, 3])
def test_segment_pages_host_pipeline_equals_page_by_page(channels):
    """sbbseg_segment_pages (groups of pages: staging / upload / compute / download overlapped on three streams) returns what
    page-by-page sbbseg_segment_page calls return (= do_prediction(True, img, model) per page, main.py:231-366) -- with a
    page count that is not a multiple of the group size, and on a second call that reuses the staging slots."""
    from sbb_textline_detection_amd.model import SegModel
    from sbb_textline_detection_amd.synthetic import synthetic_page
    from tools.synth_model import calibrated_model
    cfg, w = calibrated_model(2, 224, 224, seed=0)
    m = SegModel(cfg, w, device=0, max_batch=40, precision="f16")
    pages = [synthetic_page(500, 640, seed=30 + k) for k in range(7)]      # 12 tiles per page -> groups of 3, 3, 1
    want = [m.ctx.segment_page(p_, channels=channels) for p_ in pages]
    for rep in range(2):
        got = m.ctx.segment_pages(pages if rep == 0 else pages[::-1], channels=channels)
        ref = want if rep == 0 else want[::-1]
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            assert a.shape == b.shape and np.array_equal(a, b)
    assert np.array_equal(m.ctx.segment_pages(pages[:1], channels=channels)[0], want[0])      # a single page (one group)
    if channels == 3:                                       # the seam: a list of pages in, the reference's return layout out
        from sbb_textline_detection_amd import do_prediction, do_prediction_pages
        outs = do_prediction_pages(pages[:4], m)
        assert all(np.array_equal(o, do_prediction(True, p_, m)) for o, p_ in zip(outs, pages[:4]))
        mixed = do_prediction_pages([pages[0], pages[1][:400]], m)          # different sizes: one call per page
        assert mixed[1].shape == (400, 640, 3)
    m.release()


# ----------------------------------------------------------------------------- full-size properties (BASELINE configs[1])
@pytest.mark.parametrize("precision", ["f16x3", "f16"])
def test_full_size_pages_do_not_depend_on_chunking(precision):
    """Four 3500x2500 pages (280 tiles of 448x448, BASELINE configs[1]) through the pooled path: the label maps are the same
    bytes whether the tiles run as ONE 280-tile chunk on two lanes of 140 (the bench's launch shape), as 70-tile chunks, or page by
    page; running twice changes nothing; and every page equals its own single-page call (a patch's result does not depend on its
    batch neighbours).  f16x3 = the mode the bench number is quoted on."""
    import torch
    from sbb_textline_detection_amd.model import SegModel
    from sbb_textline_detection_amd.synthetic import synthetic_page
    from tools.synth_model import calibrated_model
    cfg, w = calibrated_model(2, 448, 448, seed=0)
    pages = [torch.from_numpy(synthetic_page(3500, 2500, seed=70 + k)).cuda() for k in range(4)]
    crcs = {}
    for mb in (280, 70):
        m = SegModel(cfg, w, device=0, max_batch=mb, precision=precision)
        outs = [torch.empty((3500, 2500), dtype=torch.uint8, device="cuda") for _ in pages]
        for rep in range(2):
            for o in outs:
                o.fill_(7)                                          # a stale buffer cannot pass for a result
            m.ctx.segment_pages_dev([p_.data_ptr() for p_ in pages], 3500, 2500, [o.data_ptr() for o in outs])
            torch.cuda.synchronize()
            crcs[(mb, rep)] = tuple(zlib.crc32(o.cpu().numpy().tobytes()) for o in outs)
        singles = []
        for p_ in pages:                                            # page by page: 70-tile launches, two lanes of 35
            single = torch.full((3500, 2500), 7, dtype=torch.uint8, device="cuda")
            m.ctx.segment_page_dev(p_.data_ptr(), 3500, 2500, single.data_ptr())
            torch.cuda.synchronize()
            singles.append(zlib.crc32(single.cpu().numpy().tobytes()))
        crcs[(mb, "page by page")] = tuple(singles)
        hist = np.bincount(outs[0].cpu().numpy().reshape(-1), minlength=2)
        assert hist[0] > 0 and hist[1] > 0 and hist[2:].sum() == 0      # a non-trivial label map, every pixel written
        m.release()
    assert len(set(crcs.values())) == 1, crcs


def test_headline_configuration_matches_oracle():
    """The bench's own configuration (bench.py, round 6), checked against the oracle: f16x3, max_batch 320, 32 pooled 3500x2500 pages (eight
    distinct ones, cycled) = 2 240 tiles = SEVEN chunks of 320 = two lanes of 160 that fork once per range, owned-region decoder launches.
    Sampled tiles from both lanes, from the first chunk and from chunks after it (the fork-once path): the page map inside a tile's
    owned region == argmax of the oracle's softmax for that tile wherever the oracle's top-2 margin exceeds EXACT_MARGIN."""
    import torch
    from oracle.keras_config import read_model_config
    from sbb_textline_detection_amd.model import SegModel
    from tools.synth_model import calibrated_model
    cfg, w = calibrated_model(2, 448, 448, seed=0)
    g = read_model_config(cfg)
    host_pages = [synthetic_page(3500, 2500, seed=k) for k in range(8)]
    dev_pages = [torch.from_numpy(p_).cuda() for p_ in host_pages]
    n_pages = 32
    m = SegModel(cfg, w, device=0, max_batch=320, precision="f16x3")
    assert m.ctx.owned_region_info() == (1, 5)
    outs = torch.full((n_pages, 3500, 2500), 9, dtype=torch.uint8, device="cuda")
    before = m.ctx.forwards()
    m.ctx.segment_pages_dev([dev_pages[k % 8].data_ptr() for k in range(n_pages)], 3500, 2500, [outs[k].data_ptr() for k in range(n_pages)])
    torch.cuda.synchronize()
    assert m.ctx.forwards() - before == n_pages * 70
    tiles, nxf, nyf = tiling.tile_grid(3500, 2500, 448, 448)
    own_map = tiling.owner_map(3500, 2500, 448, 448)
    differing = checked = 0
    seen = set()
    for pg, k in ((0, 0), (2, 69), (4, 50), (15, 31), (20, 44), (31, 69)):
        gidx = pg * 70 + k
        seen.add((gidx // 320 > 0, (gidx % 320) >= 160))
        a = outs[pg].cpu().numpy()
        assert a.max() <= 1
        assert np.array_equal(a, outs[pg % 8].cpu().numpy())                   # the same page in another chunk / lane: the same map
        t = tiles[k]
        x = (host_pages[pg % 8][t["y0"]:t["y0"] + 448, t["x0"]:t["x0"] + 448][None] / 255.0).astype(np.float32)
        ref = kf.forward(g, w, x)
        ys, xs = slice(t["y0"] + t["ylo"], t["y0"] + t["yhi"]), slice(t["x0"] + t["xlo"], t["x0"] + t["xhi"])
        own = own_map[ys, xs] == k
        r = ref[0, t["ylo"]:t["yhi"], t["xlo"]:t["xhi"]]
        srt = np.sort(r, axis=-1)
        decided = (srt[..., -1] - srt[..., -2]) > EXACT_MARGIN
        diff = (a[ys, xs] != r.argmax(-1)) & own
        differing += int(diff.sum())
        checked += int(own.sum())
        assert not (diff & decided).any(), f"page {pg} tile {k}: {int((diff & decided).sum())} labels differ from the oracle outside its near-ties"
    assert seen == {(False, False), (False, True), (True, False), (True, True)}       # first / later chunks x lane 0 / lane 1
    assert differing <= 1e-4 * checked, (differing, checked)        # measured 2.7e-5: reassociation noise at oracle margins <= 7e-5
    m.release()


def test_c_abi_rccl_collective_world1(torch_cuda, stitch_model):
    """The sharded path's collective inside the C ABI (sbbseg_comm_* / sbbseg_allgather_labels_dev: RCCL, dlopen'ed): a world of
    one on this box -- unique id, communicator, all-gather of a tile-label slice on the handle's stream, stitch -- equals the
    single-call fused path.  (World sizes > 1 need one GPU per rank: covered by the driver's multi-GPU run, not here.)"""
    torch = torch_cuda
    m = stitch_model
    ctx = m.ctx
    uid = _capi.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.comm_init(0, 1, uid)
    assert ctx.comm_info() == (0, 1)
    page = synthetic_page(900, 800, seed=12)
    xy = _capi.tile_grid(900, 800, 448, 448)[0]
    n = xy.shape[0]
    d_page = torch.from_numpy(page).cuda()
    mine = torch.empty((n, 448, 448), dtype=torch.uint8, device="cuda")
    everything = torch.zeros_like(mine)
    out = torch.empty((900, 800), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.segment_tile_range_dev(d_page.data_ptr(), 900, 800, 0, n, mine.data_ptr())
    ctx.allgather_labels_dev(mine.data_ptr(), mine.numel(), everything.data_ptr())
    ctx.stitch_dev(everything.data_ptr(), 900, 800, out.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(mine, everything)
    assert np.array_equal(out.cpu().numpy(), m.segment_page(page))
    ctx.comm_destroy()
    assert ctx.comm_info() == (0, 0)
    with pytest.raises(RuntimeError, match="no communicator"):
        ctx.allgather_labels_dev(mine.data_ptr(), mine.numel(), everything.data_ptr())
    ctx.set_stream(-1)


def test_bench_batch64_workload_on_one_gpu():
    """`bench.py --workload batch64` (BASELINE configs[3]: pages of 4000x3000 sharded as whole pages, the N > 1 default) on ONE GPU,
    cut down to two pages: the JSON line must be complete and consistent (a world of one: no exchange, every tile counted)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "batch64", "--batch-pages", "2", "--steps", "2",
                          "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--no-second-mode", "--no-extras"],
                         capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["dtype"] == "f16x3" and d["scaling"] == "strong" and d["unit"] == "patches/s"
    # 4000 % 360 = 40: the reference's call list has 12 x 9 = 108 entries per page of which one row of 9 repeats its neighbour's
    # origin (main.py:276-281); the fused path runs 11 x 9 = 99 forwards (sbbseg_set_dedupe) -- the line counts what ran
    cfgd = d["config"]
    assert cfgd["workload_id"] == "batch64" and cfgd["max_batch"] == 216
    assert cfgd["forwards_per_step"] == 2 * 99 and cfgd["tiles_per_step"] == 2 * 99 and cfgd["reference_calls_per_step"] == 2 * 108 and cfgd["dedupe"] is True
    assert d["value"] > 500 and abs(d["value"] - d["config"]["tiles_per_step"] * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 0.01 * d["value"]
    assert d["exchange"] is None and d["ranks_seen"] is None
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and r["peak"] == 2500.0 and "frac_of_split_peak" in r


def test_bench_label_match_reads_the_timed_buffer():
    """The bench line's `label_match` must be evidence about the TIMED work: it compares the label buffer the timed
    sbbseg_segment_pages_dev steps filled (page 0) with the oracle's argmax on the sampled tiles' owned regions, and the run fails
    (rc != 0) when a label differs outside EXACT_MARGIN.  Cut down: 4 pages per step, 2 sampled tiles, no second mode."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--pages-per-step", "4", "--steps", "2", "--warmup", "1",
                          "--repeats", "1", "--cpu-patches", "2", "--no-second-mode", "--no-extras"],
                         capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    lm = d["label_match"]
    assert lm["source"] == "timed_output" and lm["patches"] == 2
    assert lm["pixels_checked"] > 2 * 300 * 300 and lm["label_mismatches_outside_exact_margin"] == 0
    assert lm["label_mismatch_frac"] <= 1e-4 and lm["max_abs_softmax_diff"] < TOL_SOFTMAX["f16x3"]
    assert d["config"]["workload_id"] == "page" and d["scaling"] == "weak" and "label_check_failed" not in d
    # the driver's record keeps a bounded tail of the line: numbers only, and the label check of the timed output rides inside `roofline`
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    assert len(line) < 6000, len(line)
    assert d["roofline"]["modes"]["f16x3"]["label_mismatches_outside_exact_margin"] == 0
    cfgd = d["config"]
    assert 0.7 < cfgd["executed_share"] < 0.95 and abs(cfgd["executed_flops_per_patch"] - cfgd["executed_share"] * cfgd["flops_per_patch"]) < 1e-3 * cfgd["flops_per_patch"]
    assert d["achieved_tflops_end_to_end"] < d["reference_formulation_tflops_end_to_end"]


def test_extract_page_box_dev_equals_host_entry(torch_cuda, stitch_model):
    """sbbseg_extract_page_box_dev (the page already on the device, the border mask optional and device-side) == sbbseg_extract_page_box
    (host page in, mask out): same box, same pixel count, same mask bytes; want_mask=False changes nothing but the download."""
    torch = torch_cuda
    from sbb_textline_detection_amd.stages import scaled_size
    ctx = stitch_model.ctx
    for seed, (h, w) in enumerate([(900, 700), (1400, 1000)]):
        page = synthetic_page(h, w, seed=40 + seed)
        hs, ws = scaled_size(h, w)
        mask_h, box_h, px_h = ctx.extract_page_box(page, hs, ws)
        none, box_n, px_n = ctx.extract_page_box(page, hs, ws, want_mask=False)
        assert none is None and box_n == box_h and px_n == px_h
        d_page = torch.from_numpy(page).cuda()
        d_mask = torch.full((hs, ws), 9, dtype=torch.uint8, device="cuda")
        box_d, px_d = ctx.extract_page_box_dev(d_page.data_ptr(), h, w, hs, ws, d_mask.data_ptr())
        torch.cuda.synchronize()
        assert box_d == box_h and px_d == px_h
        assert np.array_equal(d_mask.cpu().numpy(), mask_h)
        assert ctx.extract_page_box_dev(d_page.data_ptr(), h, w, hs, ws) == (box_h, px_h)


def test_fused_paths_compute_repeated_clamped_tiles_once():
    """when the inward clamp (main.py:276-281) gives the last two tiles of an axis the same origin the reference
    runs the same forward twice.  The fused page paths skip the repeat: same label map as the reference loop (oracle tiling code
    over the HIP model's predict), fewer patches through the plan; sbbseg_set_dedupe(0) restores the reference's call count."""
    cfg, w, g, model = make_model(2, 224, 224, seed=5, precision="f16", max_batch=20)
    mid = 224 - 2 * 22                                             # 180
    # (page h, page w, distinct tiles, reference calls): repeat in y, in x, in both, in neither, and the one-tile page
    cases = [(2 * mid + 30, 3 * mid + 100, 2 * 4, 3 * 4), (3 * mid + 100, 2 * mid + 44, 4 * 2, 4 * 3), (224, 2 * mid + 7, 1 * 2, 2 * 3),
             (2 * mid + 45, 2 * mid + 45, 3 * 3, 3 * 3), (224, 224, 1, 4)]
    for hp, wp, distinct, calls in cases:
        page = synthetic_page(hp, wp, seed=hp + 3 * wp)
        xy, nx, ny = _capi.tile_grid(hp, wp, 224, 224)
        assert nx * ny == calls and len({tuple(o) for o in xy.tolist()}) == distinct, (hp, wp)
        loop = tiling.do_prediction(True, page, model)            # the reference loop: one predict per call, repeats included
        model.ctx.set_dedupe(True)
        f0 = model.ctx.forwards()
        fused = predict.do_prediction(True, page, model)
        assert model.ctx.forwards() - f0 == distinct, (hp, wp)
        model.ctx.set_dedupe(False)
        f0 = model.ctx.forwards()
        plain = predict.do_prediction(True, page, model)
        assert model.ctx.forwards() - f0 == calls, (hp, wp)
        assert np.array_equal(fused, loop) and np.array_equal(plain, loop), (hp, wp)
    # pooled pages and the crop entry point index the same smaller grid
    model.ctx.set_dedupe(True)
    hp, wp = 2 * mid + 30, 3 * mid + 100
    pages = [synthetic_page(hp, wp, seed=s) for s in (1, 2, 3)]
    f0 = model.ctx.forwards()
    pooled = model.ctx.segment_pages(pages)
    assert model.ctx.forwards() - f0 == 3 * 8
    for pg, got in zip(pages, pooled):
        assert np.array_equal(got, model.segment_page(pg))
    big = synthetic_page(hp + 60, wp + 40, seed=9)
    box = (17, 23, wp, hp)
    got = model.ctx.segment_crop(big, hp + 60, wp + 40, box, False)
    want = model.segment_page(np.ascontiguousarray(big[23:23 + hp, 17:17 + wp]))
    assert np.array_equal(got[0], want)
    model.release()


@pytest.mark.parametrize("precision", ["f16x3", "f16"])
def test_whole_image_branch_split_k_matches_oracle_and_the_unsplit_launches(precision):
    """The whole-image branch (main.py:368-380: one forward per page) runs its long-K convs split over the idle CUs (split-K, fp32
    partial sums added in split order).  Checked at the model size the stages use (448, split mode): labels against the fp32 oracle
    under the label-exact margin rule, against a handle with sbbseg_set_ksplit(0) (equal except at near-ties), bit-identical
    across repeats, and seam 2 (`predict` of the same single patch) is NOT split: it still equals the unsplit launches bit for bit."""
    from sbb_textline_detection_amd.model import SegModel
    from tools.synth_model import calibrated_model
    cfg, w = calibrated_model(2, 448, 448, seed=3)
    page = synthetic_page(1100, 900, seed=21)
    om = kf.OracleModel(cfg, w)
    x = tiling.resize_nearest(page / 255.0, 448, 448)[None].astype(np.float32)
    pr = om.predict(x)[0]
    ref = np.argmax(pr, axis=-1).astype(np.uint8)
    srt = np.sort(pr, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    split = SegModel(cfg, w, device=0, max_batch=2, precision=precision)
    f0 = split.ctx.forwards()
    got = split.segment_whole(page, 448, 448)
    again = split.segment_whole(page, 448, 448)
    assert split.ctx.forwards() - f0 == 2
    assert np.array_equal(got, again)
    # split mode: the label-exact margin rule; plain fp16: labels may differ where the oracle's margin is inside twice the mode's softmax tolerance
    lim = EXACT_MARGIN if precision == "f16x3" else 2 * TOL_SOFTMAX["f16"]
    mism = got != ref
    assert not (mism & (margin > lim)).any(), int((mism & (margin > lim)).sum())
    p_split = split.predict(x)
    plain = SegModel(cfg, w, device=0, max_batch=2, precision=precision)
    plain.ctx.set_ksplit(False)
    base = plain.segment_whole(page, 448, 448)
    differ = got != base
    assert differ.mean() < (1e-4 if precision == "f16x3" else 2e-3) and not (differ & (margin > lim)).any()
    assert np.array_equal(p_split, plain.predict(x))           # seam 2 is batch-size independent: never split
    assert np.abs(p_split[0] - pr).max() < TOL_SOFTMAX[precision]
    split.release()
    plain.release()


def test_run_with_the_page_resident_equals_the_stage_by_stage_run(tmp_path, monkeypatch):
    """InferenceStages.run() uploads the stored page once and keeps it (and the border mask, and the region map) in device memory for
    all three stages (`_run_resident`, the `_dev` entry points); SBBSEG_STAGES_RESIDENT=0 runs the stages one by one through the
    host entry points as before.  Same masks, same box, same Otsu threshold; and the 'no text' / 'crop too small' fallbacks agree."""
    from sbb_textline_detection_amd import clear_session, stages
    from sbb_textline_detection_amd.weights import save_sbbw
    from tools.synth_model import calibrated_model
    specs = {"model_page_mixed_best": 2, "model_strukturerkennung": 4, "model_textline_new": 2}      # main.py:58-60
    for name, classes in specs.items():
        cfg, w = calibrated_model(classes, 224, 224, seed=classes)
        save_sbbw(str(tmp_path / (name + ".sbbw")), cfg, w)
    st = stages.InferenceStages(*[str(tmp_path / (n + ".h5")) for n in specs], model_kwargs={"max_batch": 16})
    for hw, seed in (((520, 400), 9), ((700, 610), 4)):
        page = synthetic_page(*hw, seed=seed)
        monkeypatch.setenv("SBBSEG_STAGES_RESIDENT", "1")
        a = st.run(page)
        box_a, thr_a = st.page_box, st.otsu_threshold
        monkeypatch.setenv("SBBSEG_STAGES_RESIDENT", "0")
        b = st.run(page)
        assert box_a == st.page_box and thr_a == st.otsu_threshold and a[3] == b[3]
        for x, y in zip(a[:3], b[:3]):
            assert (x is None) == (y is None)
            if x is not None:
                assert x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x, y)
    monkeypatch.setenv("SBBSEG_STAGES_RESIDENT", "1")
    assert st._run_resident() is not None                               # (the resident path really ran above: it applies here)
    clear_session()


def test_run_is_torch_free_at_full_size(tmp_path):
    """run() (main.py:2056-2107) at BASELINE configs[2]'s size -- three 448 x 448 nets, a 3500 x 2500 page upscaled to 4200 x 3000 --
    in a child interpreter where `import torch` FAILS: the page, the border mask and the region map live in buffers the library
    allocates (sbbseg_device_alloc / _upload / _download_labels).  Resident run == stage-by-stage run there (masks, box, threshold),
    for the normal page, for a layout model that never answers class 1 (the textline model is skipped: main.py:2083, 2096) and for
    a border model whose page box is smaller than the layout model's input (the layout stage fails like main.py:278-285 ->
    regions None, main.py:2089-2091).  The parent then checks the child's maps against the oracle on sampled tiles."""
    import subprocess
    import sys
    from oracle import stage_glue
    from sbb_textline_detection_amd import clear_session
    from sbb_textline_detection_amd.keras_graph import parse_model_config
    from sbb_textline_detection_amd.model import SegModel
    from sbb_textline_detection_amd.predict import resize_nearest
    from sbb_textline_detection_amd.weights import save_sbbw
    from tools.synth_model import calibrated_model
    specs = {"model_page_mixed_best": (2, 21), "model_strukturerkennung": (4, 22), "model_textline_new": (2, 23)}      # main.py:58-60
    models = {name: calibrated_model(classes, 448, 448, seed=seed) for name, (classes, seed) in specs.items()}
    page = synthetic_page(3500, 2500, seed=33)

    def last_bn(cfg):
        return [n.name for n in parse_model_config(cfg).nodes if n.op == "bn"][-1]
    for scen in ("a", "notext", "smallbox"):
        os.makedirs(tmp_path / scen)
        for name, (cfg, w) in models.items():
            w = dict(w)
            if scen == "notext" and name == "model_strukturerkennung":
                b = w[last_bn(cfg) + "/beta:0"].copy()
                b[1] -= 60.0                                           # class 1 never wins the argmax
                w[last_bn(cfg) + "/beta:0"] = b
            if scen == "smallbox" and name == "model_page_mixed_best":
                # shift class 1's logit so that only the three most page-like pixels of the 448 x 448 border map keep it: their
                # blobs on the 4200 x 3000 page (about 9 x 7 pixels each, 25 x 25 more from the six dilations) stay below 448 x 448
                m = SegModel(cfg, w, device=0, max_batch=1)
                x = resize_nearest(resize_nearest(page, 4200, 3000), 448, 448)[None].astype(np.float32) / np.float32(255.0)
                p = m.predict(x)[0].astype(np.float64)
                m.release()
                margin = np.sort((np.log(p[..., 1]) - np.log(p[..., 0])).reshape(-1))
                b = w[last_bn(cfg) + "/beta:0"].copy()
                b[1] -= 0.5 * (margin[-3] + margin[-4])
                w[last_bn(cfg) + "/beta:0"] = b
            save_sbbw(str(tmp_path / scen / (name + ".sbbw")), cfg, w)
    clear_session()
    out = str(tmp_path / "child.npz")
    child = os.path.join(os.path.dirname(__file__), "torchfree_child.py")
    r = subprocess.run([sys.executable, child, str(tmp_path), out], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("TORCHFREE ")][-1]
    summary = json.loads(line[len("TORCHFREE "):])
    print(summary)
    assert summary["torch_loaded"] is False
    assert summary["a"]["regions"] and summary["a"]["textlines"]
    assert summary["notext"]["regions"] and not summary["notext"]["textlines"] and summary["notext"]["class1_pixels"] == 0
    assert not summary["smallbox"]["regions"] and not summary["smallbox"]["textlines"]
    assert summary["smallbox"]["box"][2] < 448 or summary["smallbox"]["box"][3] < 448
    # the child's maps against the oracle: glue bit for bit, forwards on sampled tiles (owned pixels, outside EXACT_MARGIN)
    z = np.load(out)
    bx, by, bw, bh = (int(v) for v in z["box"])
    mask3 = np.repeat(z["mask"][:, :, None], 3, axis=2)
    assert tuple(stage_glue.page_box(mask3)[0]) == (bx, by, bw, bh) and bw >= 448 and bh >= 448
    crop = resize_nearest(page, 4200, 3000)[by:by + bh, bx:bx + bw]
    assert int(z["thr"]) == stage_glue.otsu_threshold(crop[:, :, 0])
    assert stage_glue.text_regions_present(z["regions"])                    # the gate the child passed (main.py:2096)
    tiles, nxf, nyf = tiling.tile_grid(bh, bw, 448, 448)
    own = tiling.owner_map(bh, bw, 448, 448)
    k = len(tiles) // 2 + 1
    t = tiles[k]
    cfg, w = models["model_textline_new"]
    x = (crop[t["y0"]:t["y0"] + 448, t["x0"]:t["x0"] + 448][None] / 255.0).astype(np.float32)
    ref = kf.forward_config(cfg, w, x)[0]
    sl = (slice(t["y0"] + t["ylo"], t["y0"] + t["yhi"]), slice(t["x0"] + t["xlo"], t["x0"] + t["xhi"]))
    r_ = ref[t["ylo"]:t["yhi"], t["xlo"]:t["xhi"]]
    srt = np.sort(r_, axis=-1)
    mism = (z["lines"][sl] != r_.argmax(-1)) & (own[sl] == k)
    assert not (mism & ((srt[..., -1] - srt[..., -2]) > EXACT_MARGIN)).any() and mism.mean() < 1e-3
    # (the raw layout map's parity with the oracle is test_full_size_three_model_pipeline_config3's subject)


def test_run_page_one_call_equals_the_piecewise_device_calls(tmp_path):
    """sbbseg_run_page (one call: upload, border + box, layout on the Otsu'd crop + erode x 3 / dilate x 4, the text-region gate, textline)
    == the same chain spelt out with the `_dev` entry points on library-owned buffers (sbbseg_device_alloc / _upload / _download_labels,
    what INTEGRATION.md shows an integrator without torch), on two small pages with 224-pixel models."""
    from sbb_textline_detection_amd import clear_session
    from sbb_textline_detection_amd.model import load_model
    from sbb_textline_detection_amd.stages import scaled_size
    from sbb_textline_detection_amd.weights import save_sbbw
    from tools.synth_model import calibrated_model
    specs = {"model_page_mixed_best": 2, "model_strukturerkennung": 4, "model_textline_new": 2}      # main.py:58-60
    ms = []
    for name, classes in specs.items():
        cfg, w = calibrated_model(classes, 224, 224, seed=classes + 30)
        save_sbbw(str(tmp_path / (name + ".sbbw")), cfg, w)
        ms.append(load_model(str(tmp_path / (name + ".h5")), max_batch=16))
    cb, cl, ct = (m.ctx for m in ms)
    for hw, seed in (((520, 400), 9), ((700, 610), 4)):
        page = synthetic_page(*hw, seed=seed)
        H, W = hw
        Hs, Ws = scaled_size(H, W)
        mask, regions, lines, info = _capi.run_page(cb, cl, ct, page, Hs, Ws, channels=3)
        x, y, w, h = (int(v) for v in info.box_xywh)
        bufs = []

        def alloc(n):
            bufs.append(cb.device_alloc(n))
            return bufs[-1]
        try:
            d_page, d_mask = alloc(H * W * 3), alloc(Hs * Ws)
            cb.upload(d_page, page)
            box, pixels = cb.extract_page_box_dev(d_page, H, W, Hs, Ws, d_mask)
            assert box == (x, y, w, h) and pixels == info.box_pixels
            assert np.array_equal(cb.download_labels(d_mask, Hs, Ws, 3), mask)
            d_reg, d_clean, d_thr, d_lines = alloc(w * h), alloc(w * h), alloc(4), alloc(w * h)
            cl.segment_crop_dev(d_page, H, W, Hs, Ws, box, True, d_reg, d_thr)
            cl.morph_dev(d_reg, h, w, 0, 5, 3, d_clean)
            cl.morph_dev(d_clean, h, w, 1, 5, 4, d_clean)
            assert int(cl.download(d_thr, (1,), np.int32)[0]) == info.otsu_threshold and info.regions_ok == 1
            assert cl.text_regions_present_dev(d_clean, h, w) == bool(info.text_present)
            assert np.array_equal(cl.download_labels(d_clean, h, w, 3), regions)
            if info.text_present:
                ct.segment_crop_dev(d_page, H, W, Hs, Ws, box, False, d_lines)
                assert info.textlines_ok == 1 and np.array_equal(ct.download_labels(d_lines, h, w, 1), lines)
            else:
                assert lines is None
        finally:
            for b in bufs:
                cb.device_free(b)
    with pytest.raises(RuntimeError, match="not allocated by this handle"):
        cl.device_free(cb.device_alloc(64))
    clear_session()


def test_probe_switches_do_nothing_in_the_shipped_library():
    """SBBSEG_CONV_PROBE_LOCAL / _WHOT, SBBSEG_BLOCK_DBG, SBBSEG_ER_DBG made round 5's library return WRONG labels with rc 0 (timing probes).
    They compile only under -DSBBSEG_PROBES now: a process that sets all of them gets the same label map, byte for byte, in both modes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, zlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from gpu_common import make_model\n"
            "from sbb_textline_detection_amd.synthetic import synthetic_page\n"
            "page = synthetic_page(1000, 1234, seed=5)\n"
            "for prec in ('f16x3', 'f16'):\n"
            "    m = make_model(2, 448, 448, seed=0, precision=prec, max_batch=16)[3]\n"
            "    print('CRC', prec, zlib.crc32(m.segment_page(page).tobytes()) & 0xFFFFFFFF)\n"
            "    m.release()\n") % (root, os.path.join(root, "tests"))
    outs = []
    for probes in (False, True):
        env = {k: v for k, v in os.environ.items() if not k.startswith("SBBSEG_")}
        if probes:
            env.update(SBBSEG_CONV_PROBE_LOCAL="1", SBBSEG_CONV_PROBE_WHOT="1", SBBSEG_BLOCK_DBG="1", SBBSEG_ER_DBG="7")
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900, cwd=root)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append([l for l in res.stdout.splitlines() if l.startswith("CRC")])
    assert len(outs[0]) == 2 and outs[0] == outs[1], outs
