"""cutie_hnm_weights_static - the one entry point that allocates nothing (include/cutie_hip.h) - against
cutie_hnm_weights, through the C ABI: same weights and losses bit for bit, on two calls in a row on the same
GPU, the second with a smaller input (the static work area keeps the first call's histograms, and the entry
point must clear what it uses).  Also: sizes beyond the static maximum are rejected with a message."""
import numpy as np
import pytest
import torch

from cutie_amd import capi
from cutie_amd.capi import call, ptr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(n, seed, fg_share):
    g = torch.Generator().manual_seed(seed)
    ce = (torch.rand(n, generator=g) * 3.0).float()
    ce[torch.rand(n, generator=g) < 0.3] = 1.25            # many ties at one value: the tie path of the select
    labels = torch.zeros(n, dtype=torch.int32)
    fg = torch.rand(n, generator=g) < fg_share
    labels[fg] = torch.randint(1, 18, (int(fg.sum()),), generator=g, dtype=torch.int32)
    return ce.to(DEV), labels.to(DEV)


def _reference(ce, labels, ratio):
    n = labels.numel()
    w = torch.full((n,), -7.0, dtype=torch.float32, device=DEV)
    loss = torch.full((4,), -7.0, dtype=torch.float32, device=DEV)
    ws = torch.empty(capi.lib().cutie_hnm_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    call("cutie_hnm_weights", ptr(ce), ptr(labels), n, ratio, ptr(w), ptr(loss), ptr(ws), ws.numel(), capi.stream())
    return w, loss


def _static(ce, labels, ratio):
    n = labels.numel()
    w = torch.full((n,), -7.0, dtype=torch.float32, device=DEV)
    loss = torch.full((4,), -7.0, dtype=torch.float32, device=DEV)
    call("cutie_hnm_weights_static", ptr(ce), ptr(labels), n, ratio, ptr(w), ptr(loss), capi.stream())
    return w, loss


def test_static_entry_point_matches_workspace_entry_point_twice():
    torch.cuda.set_device(0)
    # first call: 32 documents of 64 x 64 cells; second call: a smaller, ragged size (not a multiple of the chunk)
    for n, seed, fg in ((32 * 64 * 64, 1, 0.05), (5 * 42 * 63 + 17, 2, 0.02), (1000, 3, 0.5)):
        ce, labels = _inputs(n, seed, fg)
        w_ref, l_ref = _reference(ce, labels, 3)
        w_got, l_got = _static(ce, labels, 3)
        torch.cuda.synchronize()
        assert np.array_equal(w_got.cpu().numpy().view(np.uint32), w_ref.cpu().numpy().view(np.uint32)), n
        assert np.array_equal(l_got[:3].cpu().numpy().view(np.uint32), l_ref[:3].cpu().numpy().view(np.uint32)), n
        assert float(l_ref[1]) > 0 and float(l_ref[2]) > 0          # the case selects something


def test_static_entry_point_rejects_oversize_and_bad_arguments():
    lib = capi.lib()
    one = torch.zeros(8, dtype=torch.float32, device=DEV)
    lab = torch.zeros(8, dtype=torch.int32, device=DEV)
    rc = lib.cutie_hnm_weights_static(ptr(one), ptr(lab), 4 * 1024 * 1024 + 1, 3, ptr(one), ptr(one), capi.stream())
    assert rc == -1 and b"exceed" in lib.cutie_last_error()
    rc = lib.cutie_hnm_weights_static(None, ptr(lab), 8, 3, ptr(one), ptr(one), capi.stream())
    assert rc == -1
