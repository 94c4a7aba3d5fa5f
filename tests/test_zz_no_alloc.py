"""ispa_nn_topk_noalloc (static device scratch, no allocation) against ispa_nn_topk, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _call_pair(torch, _abi, q, r, k, dtype_code):
    b, c, nq = q.shape
    nr = r.shape[2]
    need = _abi.lib.ispa_nn_workspace_bytes_for(dtype_code, b, c, nq, nr, k)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    d0 = torch.full((b, k, nq), -7.0, dtype=torch.float32, device="cuda")
    i0 = torch.full((b, k, nq), -7, dtype=torch.int32, device="cuda")
    d1, i1 = d0.clone(), i0.clone()
    s0 = torch.zeros(4, dtype=torch.int32, device="cuda")
    s1 = torch.zeros(4, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    _abi.check(_abi.lib.ispa_nn_topk(q.data_ptr(), r.data_ptr(), dtype_code, b, c, nq, nr, k, 1, d0.data_ptr(),
                                     i0.data_ptr(), ws.data_ptr(), need, s0.data_ptr(), stream))
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    _abi.check(_abi.lib.ispa_nn_topk_noalloc(q.data_ptr(), r.data_ptr(), dtype_code, b, c, nq, nr, k, 1,
                                             d1.data_ptr(), i1.data_ptr(), s1.data_ptr(), stream))
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.equal(i0, i1), "indices differ from ispa_nn_topk"
    assert torch.equal(d0.view(torch.int32), d1.view(torch.int32)), "distances differ from ispa_nn_topk (bitwise)"
    assert torch.equal(s0, s1)
    return d1, i1


def test_nn_topk_noalloc_matches_the_workspace_entry_point():
    import torch
    assert torch.cuda.is_available()
    from pose_estimation_amd import _abi
    from oracle import cnative
    assert _abi.lib.ispa_nn_noalloc_scratch_bytes() == 64 << 20
    g = torch.Generator(device="cuda").manual_seed(77)

    def unit(*shape):
        x = torch.randn(*shape, device="cuda", generator=g)
        return (x / x.norm(dim=1, keepdim=True)).contiguous()
    # first call: two full 64x64x128 pairs with duplicates (float64 pass) and close matches (refinement)
    q, r = unit(2, 128, 4096), unit(2, 128, 4096)
    r[:, :, 300] = r[:, :, 17]
    q[:, :, 5] = r[:, :, 17]
    q[:, :, 100:200] = r[:, :, 1000:1100] + 0.01 * torch.randn(2, 128, 100, device="cuda", generator=g)
    d, i = _call_pair(torch, _abi, q, r, 1, _abi.F32)
    want_i, want_d = cnative.nn_exact(q[1].cpu().numpy(), r[1].cpu().numpy(), 1, 1)
    assert np.array_equal(i[1].cpu().numpy(), want_i)
    assert np.all(np.abs(d[1].cpu().numpy() - want_d) <= 1e-4 * np.abs(want_d))
    # second call right after it: a smaller, ragged problem that reuses the same scratch (stale partial
    # lists and counters of the first call are still in it), bf16 and k = 5
    q2 = unit(1, 96, 264).to(torch.bfloat16)
    r2 = unit(1, 96, 520).to(torch.bfloat16)
    d2, i2 = _call_pair(torch, _abi, q2, r2, 5, _abi.BF16)
    want_i, want_d = cnative.nn_exact(q2[0].float().cpu().numpy(), r2[0].float().cpu().numpy(), 5, 1)
    assert np.array_equal(i2[0].cpu().numpy(), want_i)
    assert np.all(np.abs(d2[0].cpu().numpy() - want_d) <= 1e-4 * np.abs(want_d))
    # third: the first problem again, after the small one
    _call_pair(torch, _abi, q, r, 1, _abi.F32)
    # too large for the static buffer: a clear error, nothing launched
    big_q = torch.empty(1, dtype=torch.float32, device="cuda")
    rc = _abi.lib.ispa_nn_topk_noalloc(big_q.data_ptr(), big_q.data_ptr(), _abi.F32, 512, 128, 4096, 4096, 1, 1,
                                       big_q.data_ptr(), big_q.data_ptr(), None, None)
    assert rc == -3 and b"too large for ispa_nn_topk_noalloc" in _abi.lib.ispa_last_error()
    rc = _abi.lib.ispa_nn_topk_noalloc(big_q.data_ptr(), big_q.data_ptr(), _abi.F32, 1, 128, 4096, 4096, 9, 1,
                                       big_q.data_ptr(), big_q.data_ptr(), None, None)
    assert rc == -1


def test_nn_topk_noalloc_under_graph_capture():
    """One warm call outside capture (it looks up the scratch address and sets the kernels' launch attributes), then
    the call is captured into a HIP graph and replayed on new inputs: same bits as the eager call."""
    import torch
    from pose_estimation_amd import _abi
    g = torch.Generator(device="cuda").manual_seed(78)

    def unit(*shape):
        x = torch.randn(*shape, device="cuda", generator=g)
        return (x / x.norm(dim=1, keepdim=True)).contiguous()
    b, c, nq, nr, k = 2, 128, 1024, 2048, 1
    q, r = unit(b, c, nq), unit(b, c, nr)
    d = torch.empty((b, k, nq), dtype=torch.float32, device="cuda")
    i = torch.empty((b, k, nq), dtype=torch.int32, device="cuda")

    def call():
        _abi.check(_abi.lib.ispa_nn_topk_noalloc(q.data_ptr(), r.data_ptr(), _abi.F32, b, c, nq, nr, k, 1,
                                                 d.data_ptr(), i.data_ptr(), None,
                                                 torch.cuda.current_stream().cuda_stream))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call()                                            # warm, un-captured
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
    for _ in range(2):
        q.copy_(unit(b, c, nq))
        r.copy_(unit(b, c, nr))
        graph.replay()
        torch.cuda.synchronize()
        d_g, i_g = d.clone(), i.clone()
        call()                                            # eager, same inputs
        torch.cuda.synchronize()
        assert torch.equal(i_g, i) and torch.equal(d_g.view(torch.int32), d.view(torch.int32))
