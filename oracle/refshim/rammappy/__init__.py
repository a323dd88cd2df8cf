"""Replay stand-in for the rammappy wheel (absent here; its source is not in the reference tree).

It does NOT align anything: ``map_batch`` replays the hit records that oracle/make_golden.py registered for the
current genome in ``PENDING_HITS`` (gene index -> list of Hit).  That is enough to run the reference's own
``Serotyper.__call__`` reduction on a known hit table and record what it returns.
"""

from types import SimpleNamespace

from . import align, fasta  # noqa: F401

PENDING_HITS: dict = {}


class Preset:
    pass


class Index:
    def __init__(self, seqs):
        self.seqs = seqs

    @classmethod
    def build(cls, seqs):
        return cls(list(seqs))


class _Strand:
    def __init__(self, forward):
        self.forward = forward

    def __repr__(self):
        return "Strand.Forward" if self.forward else "Strand.Reverse"


def make_hit(target_name, target_len, q_start, q_end, t_start, t_end, strand, score, matches, block_len, mapq):
    return SimpleNamespace(
        target_name=target_name, query_start=int(q_start), query_end=int(q_end), target_len=int(target_len),
        target_start=int(t_start), target_end=int(t_end), strand=_Strand(strand > 0), block_len=int(block_len),
        matches=int(matches), edit_distance=int(block_len - matches), score=int(score), mapq=int(mapq),
        is_primary=bool(mapq > 0), is_supplementary=False, is_spliced=False,
        divergence=float((block_len - matches) / block_len) if block_len else 0.0, cs=None, md=None, cigar=b"",
    )  # fmt: skip
