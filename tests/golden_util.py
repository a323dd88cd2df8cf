"""Helpers shared by the golden-vector tests: load a typing case and rebuild its inputs."""

from __future__ import annotations

import json
from functools import lru_cache
from pathlib import Path

import numpy as np

from kaptive_amd.core.alignment import Alignments
from kaptive_amd.core.genome import GenomeAssembly
from kaptive_amd.core.seq import Sequences
from kaptive_amd.db import Database

GOLDEN = Path(__file__).resolve().parent / "golden"


def case_names() -> list[str]:
    return json.loads((GOLDEN / "typing_index.json").read_text())


# databases of the full-size cases: regenerated from their seeds (kaptive_amd.synth), not stored
SYNTH_DBS = {"kfull": ("kpsc_k", 100), "abfull": ("ab_k", 102), "ofull": ("kpsc_o", 101)}


@lru_cache(maxsize=None)
def load_db(key: str) -> Database:
    if key in SYNTH_DBS:
        from kaptive_amd.synth import make_db

        kind, seed = SYNTH_DBS[key]
        return make_db(kind, seed=seed)
    return Database.load(GOLDEN / f"db_{key}.npz")


def load_case(name: str):
    z = np.load(GOLDEN / f"typing_{name}.npz", allow_pickle=False)
    if "synth_json" in z.files:  # a full-size case: the assembly is regenerated from its seeds
        import hashlib

        from kaptive_amd.synth import make_assembly

        synth = json.loads(bytes(z["synth_json"]).decode())
        genome = make_assembly(load_db(str(z["db_key"])), name=str(z["genome_id"]), **synth["assembly"])
        assert hashlib.sha1(genome.contigs.seqs.tobytes()).hexdigest() == str(z["contig_sha1"]), "the generator has changed"
    else:
        lengths = z["contig_lengths"]
        offsets = np.zeros(len(lengths), np.int32)
        if len(lengths) > 1:
            np.cumsum(lengths[:-1], out=offsets[1:])
        contigs = Sequences(tuple(str(s) for s in z["contig_ids"]), z["contig_seqs"], offsets, lengths)
        genome = GenomeAssembly(str(z["genome_id"]), contigs)
    exp = {k[4:]: z[k] for k in z.files if k.startswith("exp.")}
    scalars = json.loads(bytes(exp.pop("scalars_json")).decode())
    kwargs = json.loads(bytes(exp.pop("typer_kwargs_json")).decode()) if "typer_kwargs_json" in exp else {}
    return str(z["db_key"]), genome, z["hits"], exp, scalars, kwargs


def hits_to_alignments(db: Database, genome: GenomeAssembly, hits: np.ndarray) -> Alignments:
    """Recorded hit table (oracle.HIT_DTYPE) -> the Alignments the native aligner would hand to the reduction."""
    if len(hits) == 0:
        return Alignments.empty()
    return Alignments.from_hit_table(
        tuple(str(i) for i in range(len(db.genes))),
        genome.contigs.ids,
        q_ids=hits["gene"], q_lengths=db.genes.lengths[hits["gene"]], q_starts=hits["q_start"],
        q_ends=hits["q_end"], t_ids=hits["contig"], t_lengths=genome.contigs.lengths[hits["contig"]],
        t_starts=hits["t_start"], t_ends=hits["t_end"], strands=hits["strand"], block_lens=hits["block_len"],
        matches=hits["matches"], scores=hits["score"], mapqs=hits["mapq"],
    )  # fmt: skip
