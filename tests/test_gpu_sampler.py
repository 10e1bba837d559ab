"""GPU parity, S rows: the HIP samplers against the oracle (= the reference's MT19937 stream) and the
golden vectors; the fast sampler against its CPU twin and the distributional contract."""
import numpy as np
import pytest

from oracle import oracle as O
from helpers import random_csr, tiny_arrays

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def test_randint_choice_known_answers(golden):
    """fresh stream, same call order as tests/golden/make_golden.py:make_sampler"""
    import skrec
    from skrec.utils.py.random import reset_global_sampler
    g = golden("golden_sampler")
    reset_global_sampler(2020)
    assert np.array_equal(skrec.randint_choice(1682, size=10, exclusion=[1, 2, 3]), g["ka1"])
    r = skrec.randint_choice(1682, size=1, exclusion=[5])
    assert np.ndim(r) == 0 and r == g["ka2"]
    assert np.array_equal(skrec.randint_choice(50, size=20, replace=False, exclusion=[0, 1, 2, 3]), g["ka3"])
    assert np.array_equal(skrec.randint_choice(30, size=15, p=g["ka4_p"]), g["ka4"])
    ka5 = skrec.batch_randint_choice(40, [3, 5, 2], exclusion=[[1, 2], [3], [4, 5, 6]], thread_num=1)
    assert np.array_equal(np.concatenate(ka5), g["ka5"])
    assert np.array_equal(skrec.randint_choice(7, size=40), g["ka6"])
    # argument checks mirror pyx_random.pyx:34-54
    with pytest.raises(ValueError):
        skrec.randint_choice(1, 3)
    with pytest.raises(ValueError):
        skrec.randint_choice(10, 0)
    with pytest.raises(TypeError):
        skrec.randint_choice(10, 3, replace=1)
    with pytest.raises(ValueError):
        skrec.randint_choice(5, 3, exclusion=[0, 1, 2, 3, 4])
    with pytest.raises(ValueError):
        skrec.randint_choice(10, 8, replace=False, exclusion=[0, 1, 2])


def test_iterators_replay_reference_epochs(golden, tiny_dir):
    """PairwiseIterator / PointwiseIterator over the tiny dataset == the reference's own output,
    continuing the known-answer sequence above on one global stream."""
    import skrec
    from skrec.io import RSDataset, PairwiseIterator, PointwiseIterator
    from skrec.utils.py.random import reset_global_sampler
    g = golden("golden_sampler")
    reset_global_sampler(2020)
    skrec.randint_choice(1682, size=10, exclusion=[1, 2, 3])
    skrec.randint_choice(1682, size=1, exclusion=[5])
    skrec.randint_choice(50, size=20, replace=False, exclusion=[0, 1, 2, 3])
    skrec.randint_choice(30, size=15, p=g["ka4_p"])
    skrec.batch_randint_choice(40, [3, 5, 2], exclusion=[[1, 2], [3], [4, 5, 6]])
    skrec.randint_choice(7, size=40)
    train = RSDataset(tiny_dir, "\t", "UIRT").train_data

    def run(it):
        cols, lens = None, []
        for batch in it:
            cols = cols or [[] for _ in batch]
            for c, b in zip(cols, batch):
                assert isinstance(b, np.ndarray)
                c.append(b)
            lens.append(len(batch[0]))
        return [np.concatenate(c, 0) for c in cols], lens
    it = PairwiseIterator(train, num_neg=1, batch_size=128, shuffle=False)
    assert len(it) == int(g["pw_len"])
    (u, i, j), lens = run(it)
    assert u.dtype == np.int32 and j.dtype == np.int32 and lens == list(g["pw_e1_lens"])
    assert np.array_equal(u, g["pw_e1_users"]) and np.array_equal(i, g["pw_e1_pos"]) and np.array_equal(j, g["pw_e1_neg"])
    (_, _, j2), _ = run(it)
    assert np.array_equal(j2, g["pw_e2_neg"])
    it3 = PairwiseIterator(train, num_neg=3, batch_size=100, shuffle=False, drop_last=True)
    assert len(it3) == int(g["pw3_len"])
    (u, i, j), lens = run(it3)
    assert j.shape == g["pw3_neg"].shape and np.array_equal(j, g["pw3_neg"]) and np.array_equal(u, g["pw3_users"])
    pt = PointwiseIterator(train, num_neg=2, batch_size=100, shuffle=False)
    assert len(pt) == int(g["pt_len"])
    (u, i, l), lens = run(pt)
    assert l.dtype == np.float32
    assert np.array_equal(u, g["pt_users"]) and np.array_equal(i, g["pt_items"]) and np.array_equal(l, g["pt_labels"])
    np.random.seed(7)
    its = PairwiseIterator(train, num_neg=1, batch_size=128, shuffle=True)
    (u, i, j), lens = run(its)
    assert np.array_equal(u, g["pws_users"]) and np.array_equal(i, g["pws_pos"]) and np.array_equal(j, g["pws_neg"])


@pytest.mark.parametrize("U,I,lo,hi,nn", [(300, 200, 0, 40, 1), (2000, 64, 1, 30, 1), (500, 1000, 0, 60, 3),
                                           (5000, 5000, 5, 80, 1), (7, 3, 1, 2, 2), (1, 50, 20, 20, 1)])
def test_exact_epoch_matches_oracle(U, I, lo, hi, nn):
    """bit-exact negatives AND identical stream position afterwards, three epochs in a row
    (several 8192-draw chunks, heavy rejection when I is small, empty rows, num_neg > 1)"""
    from gpu_utils import ExactSampler
    rng = np.random.default_rng(U * 7 + I)
    rowptr, pos = random_csr(rng, U, I, lo, hi)
    if rowptr[-1] == 0:
        pytest.skip("empty")
    ref, gpu = O.Sampler(2020), ExactSampler(2020)
    for epoch in range(3):
        want = ref.sample_epoch(I, rowptr, pos, nn).reshape(-1)
        got = gpu.epoch(I, rowptr, pos, nn)
        assert np.array_equal(got, want), (epoch, np.flatnonzero(got != want)[:5])
        assert gpu.s.draws == ref.draws
    w_ref, p_ref = ref.get_state()
    w_gpu, p_gpu = gpu.s.get_state()
    # same position in the same block, or the equivalent (block end == next block start) boundary form
    assert ref.next_u32() == _next_from_state(w_gpu, p_gpu)
    assert (p_ref % 624) == (p_gpu % 624) or {p_ref, p_gpu} <= {0, 624}


def _csr_from_lens(rng, lens, num_items):
    rowptr = np.zeros(len(lens) + 1, np.int64)
    rowptr[1:] = np.cumsum(lens)
    pos = np.concatenate([np.sort(rng.choice(num_items, l, replace=False)) for l in lens if l > 0] + [np.zeros(0, np.int64)]).astype(np.int32)
    return rowptr, pos


@pytest.mark.parametrize("case", ["uniform", "empties", "burst", "edges", "nn2"])
@pytest.mark.parametrize("path", ["slab", "serial"])
def test_exact_epoch_slab_path_matches_oracle(case, path, monkeypatch):
    """The whole-chip form of the exact epoch (slabs of 131 072 draws: detect -> resolve -> scatter, sampler.hip 2d) and
    the one-workgroup form give the oracle's negatives and stream position, bit for bit, on sparse data of every kind the
    slab path has a special answer for:
      uniform  several slabs, a few hundred events each
      empties  most users own no slot: the owner of a slot lies more than 8 users below the highest candidate -> the
               resolver hands the rest of the stream to the serial kernel
      burst    a block of users 30x denser than the data set's average: more rejections inside a slab than the window
               the detector allowed for -> hand-over in the middle of the stream
      edges    slot counts within a few draws of a slab boundary, one-slot rows, a single user
      nn2      two negatives per positive (a slot's owner is slot // 2)"""
    from gpu_utils import ExactSampler
    monkeypatch.setenv("SKR_EXACT_PATH", path)
    rng = np.random.default_rng({"uniform": 1, "empties": 2, "burst": 3, "edges": 4, "nn2": 5}[case])
    nn = 1
    if case == "uniform":
        I = 4000
        rowptr, pos = _csr_from_lens(rng, rng.integers(20, 120, 6000), I)
    elif case == "empties":
        I = 50_000
        lens = np.where(rng.random(400_000) < 0.1, rng.integers(1, 6, 400_000), 0)
        rowptr, pos = _csr_from_lens(rng, lens, I)
    elif case == "burst":
        I = 3200
        lens = np.concatenate([np.full(45_000, 2), np.full(100, 200), np.full(45_000, 2)])
        rowptr, pos = _csr_from_lens(rng, lens, I)
    elif case == "edges":
        I = 100_000
        rowptr, pos = None, None
    else:
        I, nn = 2000, 2
        rowptr, pos = _csr_from_lens(rng, rng.integers(0, 90, 3000), I)
    if case == "edges":
        for n_slots in (131072 - 2, 131072, 131072 + 1, 2 * 131072 - 40, 4096, 5000):
            lens = np.ones(n_slots, np.int64) if n_slots % 2 else np.array([n_slots])
            if len(lens) == 1:
                lens = np.array([min(n_slots, 6000)] * (n_slots // 6000) + ([n_slots % 6000] if n_slots % 6000 else []))
            rp, ps = _csr_from_lens(rng, lens, I)
            ref, gpu = O.Sampler(2020), ExactSampler(2020)
            for _ in range(2):
                assert np.array_equal(gpu.epoch(I, rp, ps, 1), ref.sample_epoch(I, rp, ps, 1)), n_slots
                assert gpu.s.draws == ref.draws
        return
    ref, gpu = O.Sampler(2020), ExactSampler(2020)
    for epoch in range(2):
        want = ref.sample_epoch(I, rowptr, pos, nn).reshape(-1)
        got = gpu.epoch(I, rowptr, pos, nn)
        assert np.array_equal(got, want), (epoch, np.flatnonzero(got != want)[:5], len(want))
        assert gpu.s.draws == ref.draws
        how = gpu.s.last_epoch()
        if path == "serial":
            assert how["status"] == 0
        else:   # the slab path ran, filled every slot, and handed over exactly where it is designed to
            assert how["status"] == 1 and how["filled"] == len(want)
            assert how["handed_over"] == (1 if case in ("empties", "burst") else 0), how
    assert ref.next_u32() == _next_from_state(*gpu.s.get_state())


def _next_from_state(words, pos):
    s = O.Sampler(1)
    s.set_state(words, pos)
    return s.next_u32()


def test_exact_sampler_state_roundtrip_and_large_epoch():
    """set_state from the middle of the oracle stream; 300k slots = 37 chunks"""
    from gpu_utils import ExactSampler
    rng = np.random.default_rng(5)
    rowptr, pos = random_csr(rng, 6000, 3000, 30, 70, empty_frac=0.02)
    ref, gpu = O.Sampler(2020), ExactSampler(99)
    for _ in range(1000):
        ref.next_u32()
    gpu.s.set_state(*ref.get_state())
    want = ref.sample_epoch(3000, rowptr, pos, 1)
    got = gpu.epoch(3000, rowptr, pos, 1)
    assert np.array_equal(got, want)
    assert gpu.s.draws == ref.draws - 1000


def test_fast_sampler_twin_and_law():
    from gpu_utils import fast_epoch
    from fast_sampler_twin import sample_fast
    rng = np.random.default_rng(11)
    rowptr, pos = random_csr(rng, 120, 40, 0, 25)
    for nn, epoch, off in ((1, 0, 0), (2, 3, 1000)):
        got = fast_epoch(2020, epoch, off, 40, rowptr, pos, nn)
        want = sample_fast(2020, epoch, off, 40, rowptr, pos, nn)
        assert np.array_equal(got, want)
    # contract: never a train positive, in range, epochs differ, shard-offset invariance
    rowptr, pos = random_csr(rng, 4000, 500, 10, 120, empty_frac=0.05)
    a = fast_epoch(1, 0, 0, 500, rowptr, pos, 1)
    b = fast_epoch(1, 1, 0, 500, rowptr, pos, 1)
    assert a.min() >= 0 and a.max() < 500 and (a != b).mean() > 0.9
    owner = np.repeat(np.arange(4000), np.diff(rowptr))
    member = np.zeros((4000, 500), bool)
    member[owner, pos] = True
    assert not member[owner, a].any()
    half = 2000
    cut = int(rowptr[half])
    lo = fast_epoch(1, 0, 0, 500, rowptr[:half + 1].copy(), pos[:cut].copy(), 1)
    hi = fast_epoch(1, 0, cut, 500, (rowptr[half:] - cut).copy(), pos[cut:].copy(), 1)
    assert np.array_equal(np.concatenate([lo, hi]), a)  # user-sharding does not change the samples
    # uniform over the allowed items: chi-square on a heavy user
    rowptr1 = np.array([0, 2000], np.int64)
    pos1 = np.sort(rng.choice(100, 20, replace=False)).astype(np.int32)
    rp = np.array([0, 20], np.int64)
    s = np.concatenate([fast_epoch(7, e, 0, 100, rp, pos1, 500) for e in range(20)])
    counts = np.bincount(s, minlength=100).astype(float)
    assert counts[pos1].sum() == 0
    allowed = np.setdiff1d(np.arange(100), pos1)
    exp = len(s) / len(allowed)
    chi2 = ((counts[allowed] - exp) ** 2 / exp).sum()
    assert chi2 < 140  # 79 dof: P(chi2 > 140) ~ 3e-5
