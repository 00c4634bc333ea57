"""The lifetime rule of pipeline batches, pinned deterministically (no repeated runs).

``DeviceBatchPipeline`` fills every batch on a WORKER stream; the consumer's kernels read it on another stream.
If the pipeline let go of a batch's tensors before the consumer's queued work on them has run, the caching
allocator would hand the same blocks to the worker's next batch (it orders reuse per stream, and these blocks
belong to the worker's stream) and the consumer would read the next batch's bytes -- the silent corruption
round 1 hit.  The rule: every tensor of a batch stays referenced until an event recorded on the consumer's
stream AFTER the consumer came back for the next batch has completed; an early close, an exception in the batch
source and a consumer on a side stream all keep it.  The tests park a long spin kernel on the consumer's stream
(torch.cuda._sleep) so that "the consumer has not finished with batch 0" is a fact, not a race."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
SPIN = 400_000_000  # GPU cycles: ~150 ms


def _pipeline(n_mols=96, batch=8, fail_at=None, workers=2):
    from deepchem_amd.data.packed_dataset import DeviceBatchPipeline
    from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
    packed = synthetic_molecules(n_mols, seed=3, max_atoms=30)
    y, w = synthetic_labels(n_mols, 2, "regression", 3)

    def index_batches():
        for k, s in enumerate(range(0, n_mols, batch)):
            if fail_at is not None and k == fail_at:
                raise RuntimeError("batch source failed")
            yield np.arange(s, s + batch), batch
    # resident=False: host collation + one H2D copy per batch on the worker stream (the path of small batches)
    return DeviceBatchPipeline(packed, y, w, index_batches(), DEV, None, depth=2, workers=workers, resident=False)


def _blocks(batch, y_t, w_t):
    return {t.untyped_storage().data_ptr() for t in (batch.graph._arena, batch.atom_features, y_t, w_t) if t is not None}


def test_blocks_of_a_batch_are_not_reissued_while_the_consumer_still_reads_it():
    pipe = _pipeline()
    it = iter(pipe)
    batch0, y0, w0 = next(it)
    first = _blocks(batch0, y0, w0)
    checksum_before = float(batch0.atom_features.sum())
    torch.cuda._sleep(SPIN)                       # the consumer's stream is busy ...
    late = batch0.atom_features.sum()             # ... and THEN reads batch 0 (queued behind the spin)
    done = torch.cuda.Event()
    done.record()
    del batch0, y0, w0                            # the consumer itself lets go, like a training loop does
    seen = []
    for k, (b, y_t, w_t) in enumerate(it):
        if not done.query():                      # the spin (and the read behind it) has not finished yet
            assert not (_blocks(b, y_t, w_t) & first), "batch %d reuses memory batch 0 is still read from" % (k + 1)
            seen.append(k)
    assert len(seen) >= 3, "the spin was too short to observe later batches being built (%s)" % seen
    torch.cuda.synchronize()
    assert float(late) == checksum_before          # what the delayed read saw is batch 0, not a later batch


def test_early_close_waits_for_the_consumer_before_releasing():
    pipe = _pipeline()
    it = iter(pipe)
    batch0, y0, w0 = next(it)
    checksum_before = float(batch0.atom_features.sum())
    torch.cuda._sleep(SPIN)
    late = batch0.atom_features.sum()
    done = torch.cuda.Event()
    done.record()
    assert not done.query()
    it.close()                                    # generator closed after one batch
    assert done.query(), "close() returned while the consumer's kernels on the last batch were still queued"
    assert float(late) == checksum_before


def test_failure_in_the_batch_source_surfaces_and_keeps_the_rule():
    pipe = _pipeline(fail_at=3)
    got = 0
    torch.cuda._sleep(SPIN // 4)
    with pytest.raises(RuntimeError, match="batch source failed"):
        for b, y_t, w_t in pipe:
            got += 1
            b.atom_features.sum()
    assert got == 3
    torch.cuda.synchronize()


def test_consumer_on_a_side_stream():
    pipe = _pipeline()
    side = torch.cuda.Stream(device=DEV)
    sums, refs = [], []
    with torch.cuda.stream(side):
        for k, (b, y_t, w_t) in enumerate(pipe):
            if k == 0:
                torch.cuda._sleep(SPIN // 2)      # on the side stream: everything below queues behind it
            sums.append(b.atom_features.sum())
            refs.append(float(0))
    side.synchronize()
    # the same batches consumed synchronously give the same sums: no batch was overwritten while queued
    ref = [float(b.atom_features.sum()) for b, _, _ in _pipeline()]
    assert [float(s) for s in sums] == ref
