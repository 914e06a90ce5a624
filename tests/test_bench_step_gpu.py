"""bench.py's train step computes the same update whether the device work is replayed from its two hipGraphs or launched eagerly --
for EVERY parameter, including those whose gradients come from stock torch ops (the conv prenet, the token embedding) and reach the
flat gradient buckets by a copy rather than through the gradient sinks of the HIP kernels."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(bench, use_graph: bool, steps: int, expect_pieces: int = 0):
    from speechain_amd import functional as Fn

    torch.manual_seed(0)
    Fn.DropoutState.reset(77)
    Fn.weights.invalidate()
    # dropout off: eager steps advance host-side mask offsets, replays advance the device seed cell, so with dropout the two modes
    # draw different masks by design and single-batch gradients differ by tens of percent (losses by 0.2 %)
    bench.ASR_MODEL = dict(bench.ASR_MODEL, dropout=0.0)
    wl = bench.ASRTrainStepWorkload(torch.device("cuda", 0), 0, 1, (64, 64))
    wl.use_graph = use_graph
    torch.manual_seed(99)  # SpecAugment's host-side draws
    losses = [float(wl.step(False)) for _ in range(steps)]
    torch.cuda.synchronize()
    assert len(wl.g_pieces) == expect_pieces, (len(wl.g_pieces), wl.segmented)
    if expect_pieces:  # every bucket is shipped exactly once, and most of them before the last piece
        shipped = sorted(i for bl in wl.piece_buckets for i in bl)
        assert shipped == list(range(len(wl.reducer.buckets))), wl.piece_buckets
        assert sum(len(bl) for bl in wl.piece_buckets[:-1]) >= len(wl.reducer.buckets) - 2, wl.piece_buckets
    # the clipped gradients of the last step, per parameter, read through the bucket slots (what the all-reduce ships and Adam uses)
    grads = {}
    coef = float(wl.opt.clip_coef) if wl.fused_opt else 1.0  # (the fused optimizer applies the clip inside its update and leaves the slots unscaled)
    for b in wl.reducer.buckets:
        for prm, view in zip(b["params"], b["views"]):
            grads[id(prm)] = view.detach().float().clone() * coef
    named = {f"{tag}.{k}": grads[id(v)] for m, tag in ((wl.prenet, "prenet"), (wl.model, "model")) for k, v in m.named_parameters()}
    in_buckets = all(prm.grad is not None and prm.grad.data_ptr() == view.data_ptr()
                     for b in wl.reducer.buckets for prm, view in zip(b["params"], b["views"]))
    del wl
    torch.cuda.empty_cache()
    return losses, named, in_buckets


def test_graph_replayed_step_equals_eager_step_for_every_parameter():
    bench = _bench()
    steps = 6  # steps 1-3 are eager in both modes; 4 captures; 5 and 6 replay
    l_e, g_e, in_e = _run(bench, False, steps)
    l_g, g_g, in_g = _run(bench, True, steps)
    assert in_e and in_g  # every param.grad is the view of its bucket slot when the step ends
    for a, b in zip(l_e, l_g):
        assert abs(a - b) <= 2e-3 * abs(a), (l_e, l_g)
    # same clipped gradient in both modes, parameter by parameter: bf16 arithmetic is deterministic here, MIOpen's conv backward is
    # not bit-reproducible, so the bound is relative to each parameter's gradient norm; a slot that was not refreshed by the replay
    # (the defect this test was written for) is off by ~100 %
    total = sum(float(v.pow(2).sum()) for v in g_e.values()) ** 0.5
    assert abs(total - 5.0) < 0.05 or total < 5.0  # the global-norm clip at 5.0 was applied to everything
    bad = {}
    for k in g_e:
        ne = float(g_e[k].norm())
        d = float((g_e[k] - g_g[k]).norm())
        if d > 5e-2 * ne + 1e-5 * total:  # 5 % of the parameter's own gradient + 1e-5 of the global norm (noise floor of near-zero ones)
            bad[k] = (d, ne)
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:8]


def test_piecewise_replayed_step_equals_eager_step():
    """The multi-rank form of the step -- forward + backward captured as FIVE hipGraphs cut at the encoder output, two hidden states and the
    encoder input, each piece's finished buckets all-reduced right after its replay -- against the eager step, parameter by parameter.
    Rehearsed on one rank with a real RCCL group and the collectives forced on (an average over one rank is the identity)."""
    import torch.distributed as dist

    bench = _bench()
    steps = 6
    l_e, g_e, in_e = _run(bench, False, steps)
    os.environ["SC_DDP_FORCE_COLLECTIVES"] = "1"
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        l_g, g_g, in_g = _run(bench, True, steps, expect_pieces=5)
    finally:
        dist.destroy_process_group()
        os.environ.pop("SC_DDP_FORCE_COLLECTIVES")
    assert in_e and in_g
    for a, b in zip(l_e, l_g):
        assert abs(a - b) <= 2e-3 * abs(a), (l_e, l_g)
    total = sum(float(v.pow(2).sum()) for v in g_e.values()) ** 0.5
    bad = {}
    for k in g_e:
        ne = float(g_e[k].norm())
        d = float((g_e[k] - g_g[k]).norm())
        if d > 5e-2 * ne + 1e-5 * total:
            bad[k] = (d, ne)
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:8]


def test_tts_replayed_step_equals_eager_step():
    """bench.py --workload tts_train_step: the FastSpeech2 step whose weight-dependent part is replayed from a hipGraph follows the same loss
    curve as the eager step (dropout off; six steps incl. the Adam updates, so a stale weight copy or an unsettled gradient would show)."""
    from speechain_amd import functional as Fn

    bench = _bench()
    bench.TTS_DROPOUT = (0.0, 0.0)
    bench.TTSTrainStepWorkload.roofline = bench.ASRTrainStepWorkload.roofline
    curves = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        Fn.DropoutState.reset(77)
        Fn.weights.invalidate()
        wl = bench.TTSTrainStepWorkload(torch.device("cuda", 0), 0, 1)
        wl.use_graph = use_graph
        curves.append([float(wl.step(False)) for _ in range(7)])
        torch.cuda.synchronize()
        assert (wl.g_body is not None) == use_graph
        del wl
        Fn.GradSinks.clear()
        torch.cuda.empty_cache()
    eager, replayed = curves
    assert eager[0] > eager[-1], eager  # it trains
    for a, b in zip(eager, replayed):
        assert abs(a - b) <= 3e-3 * abs(a), (eager, replayed)
