"""GPU parity of FeatureNormalization (HIP kernels through the C ABI) against the fixture generated from
the reference module (tests/golden/g8_featnorm.npz) and, at BASELINE sizes, against the CPU oracle.

Tolerance: the statistics are f32 sums in a different order than torch's (1e-5 relative, 1e-6 absolute); the normalised
features are one subtraction and one IEEE division on top of them."""
import pytest
import torch

from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu


def _mod(**kw):
    from speechain_amd.module.norm.feat_norm import FeatureNormalization

    return FeatureNormalization(input_size=5, **kw).cuda()


def _close(a, b, what, atol=1e-6, rtol=1e-5):
    a = a.detach().float().cpu()
    assert a.shape == b.shape, what
    assert torch.allclose(a, b, atol=atol, rtol=rtol), (what, float((a - b).abs().max()))


def _bufs(g, tag, m):
    names = {k for k, _ in m.named_buffers()}
    assert names == {k[len(tag) + 5:] for k in g if k.startswith(tag + ".buf.")}, (tag, names)
    for k, v in m.named_buffers():
        ref = g[f"{tag}.buf.{k}"]
        if v.dtype == torch.int64:
            assert torch.equal(v.cpu(), ref.to(torch.int64)), (tag, k)
        else:
            _close(v, ref, f"{tag}.{k}")


def _b(g, j):
    return g[f"feat{j}"].cuda(), g[f"len{j}"].cuda()


def test_all_norm_types_against_the_reference_fixture(golden):
    g = golden("g8_featnorm")
    x0 = g["feat0"].cuda()
    keep = x0.clone()
    _close(_mod(norm_type="utterance").train()(x0, g["len0"].cuda())[0], g["utt.y"], "utterance")
    assert torch.equal(x0, keep)  # the input is never modified
    m = _mod(norm_type="batch").train()
    _close(m(*_b(g, 0))[0], g["batch.train.y"], "batch train")
    _close(m.eval()(*_b(g, 1))[0], g["batch.eval.y"], "batch eval")
    # global = the recipes' `normalize: True`: accumulate over epochs 1, 1, 2; frozen at epoch 5 (max_epoch_num = 4); evaluation
    m = _mod().train()
    for step, (j, ep) in enumerate([(0, 1), (1, 1), (2, 2), (3, 5)]):
        _close(m(*_b(g, j), epoch=ep)[0], g[f"global.s{step}.y"], f"global step {step}")
        _bufs(g, f"global.s{step}", m)
    y = m.eval()(*_b(g, 4))[0]
    _close(y, g["global.eval.y"], "global eval")
    _close(m.recover(y), g["global.recover"], "recover", atol=1e-5)
    assert sorted(m.state_dict().keys()) == sorted(g["global.sd_keys"])
    # a checkpoint of the reference loads strictly into a fresh module (buffers are created lazily) and reproduces the evaluation
    fresh = _mod()
    fresh.load_state_dict({k: v.clone() for k, v in m.state_dict().items()}, strict=True)
    _close(fresh.eval()(*_b(g, 4))[0], g["global.eval.y"], "global eval after load_state_dict")
    m = _mod().train()  # epoch=None: the counter restarts at every call
    for step in range(2):
        _close(m(*_b(g, step))[0], g[f"global_noepoch.s{step}.y"], f"no-epoch step {step}")
        _bufs(g, f"global_noepoch.s{step}", m)
    _close(_mod(std_norm=False).train()(*_b(g, 0), epoch=1)[0], g["meanonly.y"], "mean only")
    _close(_mod(norm_type="utterance", mean_norm=False, clamp=0.5).train()(*_b(g, 0))[0], g["stdonly.y"], "std only")
    _close(_mod(norm_type="utterance").train()(g["feat0"][:, :, 0].cuda(), g["len0"].cuda())[0], g["twod.y"], "2-D input")


def test_group_normalisation_against_the_reference_fixture(golden):
    g = golden("g8_featnorm")
    m = _mod(norm_type="group").train()
    _close(m(*_b(g, 0), group_ids=7, epoch=1)[0], g["group_int.s0.y"], "group int 0")
    _close(m(*_b(g, 1), group_ids=7, epoch=1)[0], g["group_int.s1.y"], "group int 1")
    _bufs(g, "group_int.s1", m)
    m = _mod(norm_type="group").train()
    _close(m(*_b(g, 0), group_ids=g["group_ids0"].cuda(), epoch=1)[0], g["group_t.s0.y"], "group tensor 0")
    _close(m(*_b(g, 1), group_ids=g["group_ids1"].cuda(), epoch=1)[0], g["group_t.s1.y"], "group tensor 1")
    _bufs(g, "group_t.s1", m)
    _close(m.eval()(*_b(g, 2), group_ids=g["group_ids2"].cuda())[0], g["group_t.eval.y"], "unseen group -> average statistics")
    with pytest.raises(AssertionError):
        _mod(norm_type="group")(*_b(g, 0))
    with pytest.raises(ValueError):
        _mod(mean_norm=False)


def test_full_size_batch_and_degenerate_lengths_vs_oracle():
    """32 x 1001 x 80 (the frontend's output at BASELINE sizes), ragged lengths, a strided view as input; one utterance of a
    single frame (std = NaN, as torch's unbiased std) and one of two frames."""
    from speechain_amd.module.norm.feat_norm import FeatureNormalization

    gen = torch.Generator().manual_seed(8)
    B, T, F = 32, 1001, 80
    wide = torch.randn(B, T, F + 16, generator=gen) * 3 - 7
    feat = wide[:, :, 8:8 + F]  # row stride 96, not contiguous
    lens = torch.randint(500, T + 1, (B,), generator=gen)
    lens[0], lens[5], lens[6] = T, 1, 2
    for norm_type in ("utterance", "global"):
        ref_m = R.FeatureNormalization(norm_type)
        ref = ref_m(feat.contiguous(), lens, epoch=1)[0]
        m = FeatureNormalization(input_size=F, norm_type=norm_type).cuda().train()
        out = m(feat.cuda(), lens.cuda(), epoch=1)[0].cpu()
        assert out.shape == (B, T, F)
        nan_ref = torch.isnan(ref)
        assert torch.equal(torch.isnan(out), nan_ref)
        ok = ~nan_ref
        assert torch.allclose(out[ok], ref[ok], atol=2e-5, rtol=2e-5), float((out[ok] - ref[ok]).abs().max())
        if norm_type == "utterance":
            assert bool(nan_ref[5].all()) and not bool(nan_ref[6].any())  # one frame -> NaN, two frames -> finite
    # size-independent property of the utterance type: every valid region has mean 0 and unbiased std 1 per feature
    out = FeatureNormalization(input_size=F, norm_type="utterance").cuda()(feat.cuda(), lens.cuda())[0]
    for i in (0, 1, 31):
        v = out[i, : int(lens[i])]
        assert float(v.mean(0).abs().max()) < 1e-4 and float((v.std(0) - 1).abs().max()) < 1e-4


def test_refusals():
    from speechain_amd.module.norm.feat_norm import FeatureNormalization

    m = FeatureNormalization(input_size=5, norm_type="utterance").cuda()
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 4, 5), torch.tensor([4, 2]))  # CPU tensor: no fallback
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 4, 5, device="cuda", requires_grad=True), torch.tensor([4, 2], device="cuda"))
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 4, 5, 1, device="cuda"), torch.tensor([4, 2], device="cuda"))
