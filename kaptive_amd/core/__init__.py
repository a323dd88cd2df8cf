"""SoA containers used on the typing path (reference: src/kaptive/core/)."""

from kaptive_amd.core.interval import Interval, Intervals, Strand
from kaptive_amd.core.seq import SeqRecord, Sequences

__all__ = ["Interval", "Intervals", "Strand", "SeqRecord", "Sequences"]
