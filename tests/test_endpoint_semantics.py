"""Where a hit ends: kp-align's banded LOCAL alignment against an extension-with-z-drop model (what minimap2 does).

The reference's aligner (rammappy, "minimap2-based", docs/serotyping/method.md:23-28) extends a chain outwards from its
end anchors and stops each extension at the cell of maximum score (z-drop 400 only cuts extensions that have already
lost that much); kp-align runs one local alignment over the band of a seed cluster.  Hit end points feed the reduction
(coverage >= 0.2, partial / truncated states, translation frame: src/kaptive/core/alignment.py:415-446), so the VERDICT
of round 1 asked what happens at a divergent gene end.  This test states the model independently (plain numpy, no band,
anchored at the chain's first and last exact 15-mers, same scores 2 / -4, gap 4 + 2n) and compares the four end points
with the oracle's hit for genes whose last and first 60 bases diverge by 3 %, 10 % and 15 % (substitutions, and with
indels): they are identical in every case below.  The argument: both take, on each side of the anchored core, the prefix
of maximum cumulative score, and both resolve ties towards the shorter alignment (local alignment restarts at cells that
score <= 0; the extension keeps the first maximum)."""

import numpy as np
import pytest

from kaptive_amd.core.genome import GenomeAssembly
from kaptive_amd.core.seq import SeqRecord, Sequences
from kaptive_amd.synth import mutate, random_dna

MATCH, MISMATCH, GAP_O, GAP_E, ZDROP, K = 2, -4, 4, 2, 400, 15


def _extend(q: np.ndarray, t: np.ndarray) -> tuple[int, int]:
    """Extension from (0, 0): affine-gap DP over the whole rectangle, returns (rows, columns) consumed at the first cell
    of maximum score in row-major order (0, 0 when nothing scores above 0); rows whose best falls more than ZDROP below
    the running maximum end the extension."""
    n, m = len(q), len(t)
    neg = -(10**9)
    H = np.full((n + 1, m + 1), neg, np.int64)
    E = np.full((n + 1, m + 1), neg, np.int64)
    F = np.full((n + 1, m + 1), neg, np.int64)
    H[0, 0] = 0
    for j in range(1, m + 1):
        E[0, j] = max(H[0, j - 1] - GAP_O - GAP_E, E[0, j - 1] - GAP_E)
        H[0, j] = E[0, j]
    best, at = 0, (0, 0)
    for i in range(1, n + 1):
        F[i, 0] = max(H[i - 1, 0] - GAP_O - GAP_E, F[i - 1, 0] - GAP_E)
        H[i, 0] = F[i, 0]
        row_best = H[i, 0]
        for j in range(1, m + 1):
            E[i, j] = max(H[i, j - 1] - GAP_O - GAP_E, E[i, j - 1] - GAP_E)
            F[i, j] = max(H[i - 1, j] - GAP_O - GAP_E, F[i - 1, j] - GAP_E)
            H[i, j] = max(H[i - 1, j - 1] + (MATCH if q[i - 1] == t[j - 1] else MISMATCH), E[i, j], F[i, j])
            row_best = max(row_best, H[i, j])
            if H[i, j] > best:
                best, at = int(H[i, j]), (i, j)
        if best - row_best > ZDROP:
            break
    return at


def _model_ends(gene: np.ndarray, contig: np.ndarray) -> tuple[int, int, int, int]:
    """q_start, q_end, t_start, t_end of the extension model; anchors = first and last exact K-mer shared on the main
    diagonal band (the test plants the gene without large indels, so a shared K-mer within 8 diagonals is an anchor)."""
    index = {}
    for p in range(len(contig) - K + 1):
        index.setdefault(contig[p : p + K].tobytes(), []).append(p)
    anchors = [(i, p) for i in range(len(gene) - K + 1) for p in index.get(gene[i : i + K].tobytes(), ())]
    assert anchors, "no anchor: the case is outside what either aligner would report"
    (q0, t0), (q1, t1) = min(anchors), max(anchors)
    left = _extend(gene[:q0][::-1], contig[:t0][::-1])
    right = _extend(gene[q1 + K :], contig[t1 + K :])
    return q0 - left[0], q1 + K + right[0], t0 - left[1], t1 + K + right[1]


def _diverge_ends(rng, gene: np.ndarray, rate: float, indels: bool) -> np.ndarray:
    core = gene[60:-60]
    head = mutate(rng, gene[:60], rate, indel_rate=0.02 if indels else 0.0)
    tail = mutate(rng, gene[-60:], rate, indel_rate=0.02 if indels else 0.0)
    return np.concatenate([head, mutate(rng, core, 0.01), tail])


@pytest.mark.parametrize("rate", [0.03, 0.10, 0.15])
@pytest.mark.parametrize("indels", [False, True])
def test_hit_ends_equal_extension_model(oracle, rate, indels):
    rng = np.random.default_rng(int(rate * 1000) + (7 if indels else 0))
    checked = 0
    for trial in range(4):
        gene = random_dna(rng, 420, 0.5)
        copy = _diverge_ends(rng, gene, rate, indels)
        contig = np.concatenate([random_dna(rng, 150, 0.5), copy, random_dna(rng, 170, 0.5)])
        codes = np.searchsorted(np.frombuffer(b"ACGT", np.uint8), gene).astype(np.uint8)
        odb = oracle.OracleDB(codes, np.array([0, len(gene)], np.int32))
        genome = GenomeAssembly("g", Sequences.from_records([SeqRecord("c", contig.tobytes())]))
        hits = odb.align(genome.packed())
        hits = hits[hits["strand"] > 0]
        if len(hits) == 0:
            continue  # fewer than three seeds survived: not reported (minimap2's -n 3 would drop the chain as well)
        h = hits[np.argmax(hits["score"])]
        want = _model_ends(gene, contig)
        got = (int(h["q_start"]), int(h["q_end"]), int(h["t_start"]), int(h["t_end"]))
        assert got == want, (rate, indels, trial, got, want)
        checked += 1
    assert checked >= 3
