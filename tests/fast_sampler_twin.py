"""CPU twin of ``sample_fast_kernel`` (csrc/sampler.hip): slot-keyed xoshiro128++ + Lemire mapping +
rejection of the user's positives.  Vectorised numpy uint64/uint32 arithmetic; used only by tests to
check the GPU kernel bit for bit.  It is NOT a restatement of the reference (whose stream is
MT19937); the reference is matched by this sampler in law only."""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
    return x, z ^ (z >> np.uint64(31))


def _rotl(x, k):
    return ((x << np.uint32(k)) | (x >> np.uint32(32 - k))).astype(np.uint32)


def sample_fast(seed, epoch, slot_offset, num_items, rowptr, pos_sorted, num_neg):
    with np.errstate(over="ignore"):
        nnz = int(rowptr[-1])
        n_slots = nnz * num_neg
        slots = np.arange(n_slots, dtype=np.uint64) + np.uint64(slot_offset)
        x = np.full(n_slots, seed, dtype=np.uint64)
        x, k = _splitmix(x)
        k = k ^ ((np.uint64(epoch) * np.uint64(0xD1B54A32D192ED03)) & M64)
        x, k = _splitmix(k)
        k = k ^ slots
        x, a = _splitmix(k)
        x, b = _splitmix(x)
        s0 = (a & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        s1 = (a >> np.uint64(32)).astype(np.uint32)
        s2 = (b & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        s3 = (b >> np.uint64(32)).astype(np.uint32)
        zero = (s0 | s1 | s2 | s3) == 0
        s0[zero] = 1
        owner = np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr) * num_neg)
        out = np.full(n_slots, -1, np.int32)
        todo = np.arange(n_slots)
        high = np.uint64(num_items)
        thr = np.uint32((2 ** 32 - num_items) % num_items)
        for _ in range(64 * num_items + 1024):
            if len(todo) == 0:
                break
            t0, t1, t2, t3 = s0[todo], s1[todo], s2[todo], s3[todo]
            r = (_rotl((t0 + t3).astype(np.uint32), 7) + t0).astype(np.uint32)
            t = (t1 << np.uint32(9)).astype(np.uint32)
            t2 = t2 ^ t0
            t3 = t3 ^ t1
            t1 = t1 ^ t2
            t0 = t0 ^ t3
            t2 = t2 ^ t
            t3 = _rotl(t3, 11)
            s0[todo], s1[todo], s2[todo], s3[todo] = t0, t1, t2, t3
            prod = r.astype(np.uint64) * high
            low = (prod & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            cand = (prod >> np.uint64(32)).astype(np.int32)
            ok = low >= thr
            hit = np.zeros(len(todo), bool)
            for n, slot in enumerate(todo):
                if ok[n]:
                    u = owner[slot]
                    row = pos_sorted[rowptr[u]:rowptr[u + 1]]
                    p = np.searchsorted(row, cand[n])
                    hit[n] = p < len(row) and row[p] == cand[n]
            done = ok & ~hit
            out[todo[done]] = cand[done]
            todo = todo[~done]
        return out
