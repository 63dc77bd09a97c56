"""-m gpu: tfc_csr_build_noalloc, the entry point that allocates nothing (its scratch is a static device
array), called through the C ABI and compared bit for bit with tfc_csr_build on two calls in a row, the
second with a smaller input (the shared array keeps the first call's contents)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

MAX_INTS = 1 << 22  # TFC_CSR_NOALLOC_MAX_INTS (include/tfc.h)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _both(L, ei, n_nodes, key_row, stream):
    """(row_ptr, eid, nbr, status) from tfc_csr_build and from tfc_csr_build_noalloc."""
    dev = ei.device
    E = ei.shape[1]
    outs = []
    for noalloc in (False, True):
        rp = torch.full((n_nodes + 1,), -7, dtype=torch.int32, device=dev)
        eid = torch.full((max(E, 1),), -7, dtype=torch.int32, device=dev)
        nbr = torch.full((max(E, 1),), -7, dtype=torch.int32, device=dev)
        status = torch.full((1,), -7, dtype=torch.int32, device=dev)
        key, other = ei[key_row].contiguous(), ei[1 - key_row].contiguous()
        if noalloc:
            rc = L.tfc_csr_build_noalloc(_p(key), _p(other), E, n_nodes, _p(rp), _p(eid), _p(nbr), _p(status), stream)
        else:
            ws = torch.empty(max(int(L.tfc_csr_workspace_bytes(n_nodes, E)), 8), dtype=torch.uint8, device=dev)
            rc = L.tfc_csr_build(_p(key), _p(other), E, n_nodes, _p(rp), _p(eid), _p(nbr), _p(status), _p(ws), stream)
        assert rc == 0, L.tfc_last_error()
        torch.cuda.synchronize()
        outs.append((rp, eid[:E], nbr[:E], status))
    return outs


def test_csr_build_noalloc_matches_csr_build(device):
    from matten_amd import _lib

    L = _lib.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(11)
    # two calls in a row, the second smaller; then an empty one, one with bad indices, one with very dense nodes
    cases = [(3000, 90000), (50, 400), (7, 0), (5, 3000), (1, 1)]
    for n_nodes, n_edges in cases:
        ei = torch.randint(0, n_nodes, (2, n_edges), generator=g).to(device)
        for key_row in (1, 0):
            ref, got = _both(L, ei, n_nodes, key_row, stream)
            for a, b in zip(ref, got):
                assert torch.equal(a, b), (n_nodes, n_edges, key_row)
            assert int(got[3].item()) == 0
    ei = torch.tensor([[0, 1, 5, 2], [1, 0, 1, -1]], device=device)
    ref, got = _both(L, ei, 3, 1, stream)
    assert int(got[3].item()) == 1 and all(torch.equal(a, b) for a, b in zip(ref, got))


def test_csr_build_noalloc_allocates_nothing_and_rejects_oversize(device):
    from matten_amd import _lib

    L = _lib.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n_nodes, n_edges = 2000, 60000
    ei = torch.randint(0, n_nodes, (2, n_edges), generator=torch.Generator().manual_seed(3)).to(device)
    rp = torch.empty(n_nodes + 1, dtype=torch.int32, device=device)
    eid, nbr = (torch.empty(n_edges, dtype=torch.int32, device=device) for _ in range(2))
    status = torch.zeros(1, dtype=torch.int32, device=device)
    args = (_p(ei[1].contiguous()), _p(ei[0].contiguous()), n_edges, n_nodes, _p(rp), _p(eid), _p(nbr), _p(status), stream)
    assert L.tfc_csr_build_noalloc(*args) == 0  # first use on this device: the address lookup happens here
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        assert L.tfc_csr_build_noalloc(*args) == 0
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0  # device memory in use did not move
    # beyond the static scratch: refused with a clear message, nothing launched
    rp.fill_(-3)
    rc = L.tfc_csr_build_noalloc(args[0], args[1], MAX_INTS, n_nodes, *args[4:])
    assert rc == -2 and b"exceed the static scratch" in L.tfc_last_error()
    rc = L.tfc_csr_build_noalloc(args[0], args[1], MAX_INTS - n_nodes - 1, n_nodes, *args[4:])
    assert rc == -2
    torch.cuda.synchronize()
    assert bool((rp == -3).all())
    assert L.tfc_csr_build_noalloc(args[0], args[1], -1, n_nodes, *args[4:]) == -1
