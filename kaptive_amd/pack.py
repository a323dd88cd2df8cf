"""2-bit packing of contigs into the device layout described in include/kp_spec.h.

This replaces what the reference feeds its aligner -- a list of ``(name, bytes)`` per contig copied out of the
``Sequences`` container (src/kaptive/core/genome.py:188, src/kaptive/core/seq.py:281-290) -- with one padded
coordinate space per assembly: packed words, contig starts/lengths and sorted N runs.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from kaptive_amd.core.seq import CHAR_MAP, Sequences

CONTIG_ALIGN = 32  # KP_CONTIG_ALIGN
ASM_ALIGN = 64  # KP_ASM_ALIGN
_SHIFTS = (2 * np.arange(16, dtype=np.uint32)).astype(np.uint32)


def _round_up(x, m):
    return (x + m - 1) // m * m


@dataclass(frozen=True, slots=True)
class PackedAssembly:
    words: np.ndarray  # uint32, padded_len / 16 words
    padded_len: int  # bases, multiple of ASM_ALIGN
    ctg_start: np.ndarray  # int32 [C] start of each contig in the padded space
    ctg_len: np.ndarray  # int32 [C]
    n_runs: np.ndarray  # int32 [R, 2] sorted disjoint [start, end) runs of non-ACGT bases (padded space)

    @property
    def n_bases(self) -> int:
        return int(self.ctg_len.sum())


def codes_to_words(codes: np.ndarray) -> np.ndarray:
    """uint8 codes 0..3 (length multiple of 16) -> little-endian 2-bit words."""
    c = codes.reshape(-1, 16).astype(np.uint32)
    return np.bitwise_or.reduce(c << _SHIFTS, axis=1).astype(np.uint32)


def words_to_codes(words: np.ndarray, n: int | None = None) -> np.ndarray:
    c = ((words[:, None] >> _SHIFTS) & 3).astype(np.uint8).reshape(-1)
    return c if n is None else c[:n]


def pack_contigs(contigs: Sequences) -> PackedAssembly:
    lens = contigs.lengths.astype(np.int64)
    slots = _round_up(lens, CONTIG_ALIGN)
    starts = np.zeros(len(lens), np.int64)
    if len(lens) > 1:
        np.cumsum(slots[:-1], out=starts[1:])
    padded = int(_round_up(int(slots.sum()), ASM_ALIGN)) if len(lens) else 0
    if padded > (1 << 30) - 65536:
        raise ValueError("assembly too long for the packed layout (KP_MAX_ASM_LEN)")
    codes = np.zeros(padded, dtype=np.uint8)
    n_mask = np.zeros(padded + 1, dtype=bool)
    for o, n, s in zip(contigs.offsets.tolist(), lens.tolist(), starts.tolist()):
        c = CHAR_MAP[contigs.seqs[o : o + n]]
        bad = c == 4
        if bad.any():
            n_mask[s : s + n] = bad
            c = np.where(bad, 0, c)
        codes[s : s + n] = c
    edges = np.flatnonzero(n_mask[1:] != n_mask[:-1]) + 1
    if n_mask[0]:
        edges = np.r_[0, edges]
    # a run never spans two contigs (kp_fasta_ingest / kp_pack_contigs start a new one with every contig): where a contig
    # that fills its slot ends in an ambiguous base and the next one starts with one, the run is cut at the boundary
    inner = np.unique(starts[1:])  # (empty contigs share their start with the next one)
    cut = inner[n_mask[inner - 1] & n_mask[inner]] if len(inner) else inner
    if len(cut):
        edges = np.sort(np.r_[edges, cut, cut], kind="stable")
    runs = edges.reshape(-1, 2).astype(np.int32) if len(edges) else np.empty((0, 2), np.int32)
    return PackedAssembly(
        codes_to_words(codes) if padded else np.empty(0, np.uint32),
        padded,
        starts.astype(np.int32),
        lens.astype(np.int32),
        runs,
    )


def pack_sequences_flat(seqs: Sequences) -> tuple[np.ndarray, np.ndarray]:
    """Genes for the device: codes 0..4 as one byte per base (N kept as 4), plus int32 offsets (length n+1)."""
    codes = CHAR_MAP[seqs._dense_bytes()]
    offs = np.zeros(len(seqs) + 1, np.int32)
    np.cumsum(seqs.lengths, out=offs[1:])
    return codes, offs
