"""Typing engine, result containers and TSV rows (reference: src/kaptive/serotyping/)."""

from kaptive_amd.serotyping.io import KaptiveRow, Pha4geRow, ReportRow
from kaptive_amd.serotyping.models import GeneHits, GeneState, LocusPieces, SerotypingProblem, SerotypingResult

__all__ = ["GeneHits", "GeneState", "KaptiveRow", "LocusPieces", "Pha4geRow", "ReportRow", "SerotypingProblem",
           "SerotypingResult", "Serotyper"]  # fmt: skip


def __getattr__(name: str):
    if name == "Serotyper":  # imported lazily: it pulls in the native engine glue
        from kaptive_amd.serotyping.core import Serotyper

        return Serotyper
    raise AttributeError(name)
