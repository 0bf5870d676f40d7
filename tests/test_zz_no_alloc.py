"""kmers_hip_count_y_static (include/kmers_hip.h): the no-allocation variant of kmers_hip_count_y_ws, whose K >= 9
scratch is a static __device__ array inside the library.  Called through the C ABI twice in a row, the second time with
a smaller input (the array keeps the first call's contents), and compared bit for bit with kmers_hip_count_y_ws and, for the first store, with the oracle's fasta_count_kmers on the same
file.  After the warm-up call (the per-device symbol lookup) a call leaves the device's free memory as it found it."""
import numpy as np
import pytest

from kmers_amd import _lib, synth
from tests import kmtest
from kmers_amd.store import MODE_Y_FASTA, Store

pytestmark = pytest.mark.gpu


def _static(lib, ds, K, counts):
    rc = lib.kmers_hip_count_y_static(ds.bases.data_ptr(), ds.bases.numel(), ds.valid.data_ptr(), ds.valid.numel(),
                                      ds.n_pos, K, counts.data_ptr(), None)
    _lib.check(rc, "kmers_hip_count_y_static")


@pytest.mark.parametrize("K", [12, 14, 8])
def test_static_scratch_equals_workspace_entry(tmp_path, K):
    import torch
    lib = _lib.handle()
    iso = str(tmp_path / "tx.fsa")
    info = synth.synth_transcriptome(iso, n_records=3000, total_bases=6_000_000, seed=7, return_sequences=True)
    stores, paths = [], []
    for i, n_reads in enumerate((150_000, 40_000)):          # 15.2 M positions (two chunks of the 32 MiB array), then 4 M
        p = str(tmp_path / f"reads{i}.fsa")
        synth.synth_reads(p, info["concat"], info["offsets"], n_reads, read_len=100, seed=11 + i)
        stores.append(Store(p, MODE_Y_FASTA).to_device())
        paths.append(p)
    z = torch.zeros(8, dtype=torch.int32, device="cuda")
    assert lib.kmers_hip_count_y_static(None, 0, None, 0, 0, K, z.data_ptr(), None) == 0     # warm-up: the symbol lookup
    assert not z.any()
    for i, ds in enumerate(stores):                                        # two calls in a row on the same array
        want = ds.count_y(K)                                 # kmers_hip_count_y_ws with a torch-allocated workspace
        got = torch.zeros(4 ** K, dtype=torch.int32, device=ds.device)
        torch.cuda.synchronize()
        before = (torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
        _static(lib, ds, K, got)
        torch.cuda.synchronize()
        assert (torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated(), torch.cuda.memory_reserved()) == before
        assert int(want.sum()) > 0
        assert torch.equal(got, want)
        if i == 0 and K == 12:                               # and the checker itself, not only the sibling entry point
            rc, y = kmtest.load_oracle().y([paths[0]], K)
            assert rc == 0 and np.array_equal(got.cpu().numpy(), y)
    # accumulates like the other entry points
    _static(lib, stores[1], K, got)
    torch.cuda.synchronize()
    assert torch.equal(got, 2 * want)


def test_static_scratch_rejects_oversize():
    import torch
    lib = _lib.handle()
    z = torch.zeros(8, dtype=torch.int32, device="cuda")
    n_pos = (1 << 28) + 1
    rc = lib.kmers_hip_count_y_static(z.data_ptr(), (n_pos + 15) // 16, z.data_ptr(), (n_pos + 31) // 32, n_pos, 12,
                                      z.data_ptr(), None)
    assert rc == -13          # KMERS_ERR_TOO_LARGE, before anything is launched
