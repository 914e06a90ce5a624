"""Host-side logic that needs no GPU: the layout of the flat gradient / parameter buckets and the staleness rules of the weight cache."""
import pytest
import torch


def test_bucket_slots_start_on_16_byte_boundaries():
    from speechain_amd.ddp import GradBucketReducer

    ps = [torch.nn.Parameter(torch.randn(s)) for s in [(6, 5), (1,), (7,), (3, 3), (16,), (2,)]]
    q, k, v = (torch.nn.Parameter(torch.randn(3, 5)) for _ in range(3))  # 15 elements each, kept back to back
    red = GradBucketReducer(ps + [q, k, v], adjacent=[[q, k, v]], sinks=False)
    b = red.buckets[0]
    index = {id(p): i for i, p in enumerate(b["params"])}
    offs = b["offsets"]
    for p in ps + [q]:
        assert offs[index[id(p)]] % 4 == 0  # every slot (every group) starts 16-byte aligned ...
    assert offs[index[id(k)]] == offs[index[id(q)]] + 15 and offs[index[id(v)]] == offs[index[id(q)]] + 30  # ... groups stay glued
    for p, o, view in zip(b["params"], offs, b["views"]):
        assert p.grad is view and view.data_ptr() == b["flat"].data_ptr() + 4 * o
    # slots do not overlap and the padding is zero
    cover = torch.zeros_like(b["flat"])
    for p, o in zip(b["params"], offs):
        cover[o: o + p.numel()] += 1
    assert cover.max() == 1
    vals = [p.detach().clone() for p in b["params"]]
    flat = red.flatten_parameters()[0]
    for p, val, o in zip(b["params"], vals, offs):
        assert torch.equal(p.detach(), val) and p.data_ptr() == flat.data_ptr() + 4 * o
    assert float(flat[cover == 0].abs().sum()) == 0.0


def test_small_buckets_never_split_a_glued_group():
    from speechain_amd.ddp import GradBucketReducer

    q, k, v = (torch.nn.Parameter(torch.randn(40, 10)) for _ in range(3))
    other = [torch.nn.Parameter(torch.randn(300)) for _ in range(3)]
    red = GradBucketReducer(other + [q, k, v], bucket_mb=500 * 4 / 1024 / 1024, adjacent=[[q, k, v]], sinks=False)
    home = {id(p): bi for bi, b in enumerate(red.buckets) for p in b["params"]}
    assert home[id(q)] == home[id(k)] == home[id(v)]
    assert len(red.buckets) >= 3


def test_bias_vector_cache_follows_mark_stale():
    """An optimizer that writes parameters with a raw kernel leaves torch's version counters alone; `mark_stale()` is its way of saying so."""
    from speechain_amd.functional import _WeightCache

    c = _WeightCache()
    a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.ones(2))
    v0 = c.vector([a, b])
    assert v0.tolist() == [0, 0, 0, 1, 1]
    a.data.view(-1).numpy()[:] = 5.0  # an update behind the version counter
    assert c.vector([a, b]) is v0 and v0.tolist() == [0, 0, 0, 1, 1]  # not noticed: that is what the flag is for
    c.mark_stale()
    v1 = c.vector([a, b])
    assert v1.tolist() == [5, 5, 5, 1, 1]
    assert v1.data_ptr() == v0.data_ptr()  # rebuilt in place: a captured graph may hold the address
    with torch.no_grad():
        b.add_(1.0)  # an ordinary torch update moves the version counter
    assert c.vector([a, b]).tolist() == [5, 5, 5, 2, 2]
    assert c.vector([None, None]) is None
    with pytest.raises(ValueError):
        c.vector([a, None])
