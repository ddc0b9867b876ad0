"""Minimal sequence plumbing for tests and bench.py: FASTA in/out, reverse complement,
and the seeded synthetic-pair generator of SURVEY.md section 8(d).

None of this is on the hot path; the reference's own file I/O (src/sequences.c) is out of
scope.  Reverse complement follows the reference's nuc_to_complement table
(src/dna_utilities.c:96-114: IUPAC-aware, case-preserving, everything else unchanged).
"""
import numpy as np

_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip("ACGTRYKMBVDHNSW", "TGCAYRMKVBHDNSW"):
    _COMP[ord(_a)] = ord(_b)
    _COMP[ord(_a.lower())] = ord(_b.lower())


def revcomp(seq):
    a = np.frombuffer(bytes(seq), dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
    return _COMP[a[::-1]]


def complement_table():
    return _COMP.copy()


def read_fasta(path):
    """-> list of (name, uint8 array).  No soft-mask handling: bytes are kept as-is."""
    out, name, chunks = [], None, []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    out.append((name, np.frombuffer(b"".join(chunks), dtype=np.uint8).copy()))
                name, chunks = line[1:].decode(), []
            elif line:
                chunks.append(line)
    if name is not None:
        out.append((name, np.frombuffer(b"".join(chunks), dtype=np.uint8).copy()))
    return out


def write_fasta(path, records, width=60):
    with open(path, "wb") as f:
        for name, seq in records:
            f.write(b">" + name.encode() + b"\n")
            b = bytes(seq)
            for i in range(0, len(b), width * 1000):
                block = b[i:i + width * 1000]
                f.write(b"\n".join(block[j:j + width] for j in range(0, len(block), width)) + b"\n")


def _mutate(rng, block, sub_rate, indel_rate):
    """substitutions + single-base indels (half insertions, half deletions)"""
    n = len(block)
    b = block.copy()
    sub = rng.random(n) < sub_rate
    if sub.any():
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        code = np.zeros(256, dtype=np.int64)
        code[acgt] = np.arange(4)
        b[sub] = acgt[(code[b[sub]] + rng.integers(1, 4, int(sub.sum()))) % 4]
    ev = rng.random(n)
    dele = ev < indel_rate / 2
    ins = (ev >= indel_rate / 2) & (ev < indel_rate)
    if not (dele.any() or ins.any()):
        return b
    reps = np.ones(n, dtype=np.int64)
    reps[dele] = 0
    reps[ins] = 2
    out = np.repeat(b, reps)
    # the duplicated base of an insertion becomes a random base
    idx = np.cumsum(reps)[ins] - 1
    out[idx] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, len(idx))]
    return out


def synth_pair(tlen, qlen, seed=1, homolog_frac=0.5, sub_rate=0.12, indel_rate=0.01,
               block_min=2000, block_max=20000, revcomp_frac=0.5):
    """Deterministic (target, query) pair, SURVEY.md 8(d):
    target = iid uniform ACGT; query = concatenation of 2-20 kbp blocks, each with p=0.5 a
    mutated copy (12 % substitutions, 1 % indels) of a random target interval -- half of those
    reverse-complemented -- else iid filler."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    target = acgt[rng.integers(0, 4, tlen)]
    return target, _query_of(rng, target, qlen, homolog_frac, sub_rate, indel_rate, block_min, block_max, revcomp_frac)


def synth_query(target, qlen, seed, homolog_frac=0.5, sub_rate=0.12, indel_rate=0.01,
                block_min=2000, block_max=20000, revcomp_frac=0.5):
    """another query for an existing target (multi-sequence query sets: BASELINE.json configs[3]), same recipe"""
    return _query_of(np.random.default_rng(seed), target, qlen, homolog_frac, sub_rate, indel_rate, block_min, block_max, revcomp_frac)


def _query_of(rng, target, qlen, homolog_frac, sub_rate, indel_rate, block_min, block_max, revcomp_frac):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    tlen = len(target)
    parts, have = [], 0
    while have < qlen:
        blen = int(rng.integers(block_min, block_max + 1))
        blen = min(blen, qlen - have, tlen)
        if rng.random() < homolog_frac and blen >= 64:
            s = int(rng.integers(0, tlen - blen + 1))
            blk = _mutate(rng, target[s:s + blen], sub_rate, indel_rate)
            if rng.random() < revcomp_frac:
                blk = revcomp(blk)
        else:
            blk = acgt[rng.integers(0, 4, blen)]
        parts.append(blk)
        have += len(blk)
    return np.concatenate(parts)[:qlen].copy()
