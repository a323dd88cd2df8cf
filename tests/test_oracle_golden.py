"""The CPU oracle against golden vectors recorded from the reference (oracle/make_golden.py).

These pin oracle/kp_oracle.c: protein DP (pairwise.py:395-584), overlap cull and clustering (interval.py:595-751),
extract / translate (seq.py:612-741).
"""

import numpy as np


def test_protein_dp_matches_reference(oracle, golden_dir):
    z = np.load(golden_dir / "protein_dp.npz")
    got = oracle.protein_align(z["q_seqs"], z["q_offsets"], z["q_lengths"], z["t_seqs"], z["t_offsets"], z["t_lengths"])
    for i, col in enumerate(("scores", "matches", "mismatches", "gaps", "q_starts", "q_ends", "t_starts", "t_ends")):
        assert np.array_equal(got[:, i], z[col]), col
    # the known answers quoted in SURVEY.md Appendix C
    assert got[0].tolist() == [69, 14, 1, 0, 0, 15, 0, 15]
    assert got[2].tolist() == [149, 32, 0, 1, 0, 32, 0, 33]
    assert got[3].tolist() == [10, 2, 0, 0, 0, 2, 0, 2]
    assert got[7].tolist() == [26, 6, 1, 0, 0, 7, 0, 7]


def test_blosum_table_spot_values(oracle):
    m = oracle.blosum62()
    assert m[ord("W"), ord("W")] == 11 and m[ord("A"), ord("R")] == -1 and m[ord("*"), ord("*")] == 1
    assert m[ord("a"), ord("A")] == -128 and m[ord("X"), ord("X")] == -1 and m[ord("B"), ord("D")] == 4


def test_interval_reductions_match_reference(oracle, golden_dir):
    z = np.load(golden_dir / "intervals.npz")
    for i in range(int(z["n_cases"])):
        s, e, g, order = (z[f"c{i}_{k}"] for k in ("starts", "ends", "groups", "order"))
        kept = oracle.cull_overlaps(order, g, np.zeros_like(g), s, e, 0.1)
        assert np.array_equal(kept, z[f"c{i}_kept"]), i
        corder = np.lexsort((e, s, g)).astype(np.int32)
        ids = oracle.cluster(s, e, g, int(z[f"c{i}_tol"]), corder)
        assert np.array_equal(ids, z[f"c{i}_clusters"]), i
    assert z["c0_kept"].tolist() == [True, False, True, True, True]
    assert z["c0_clusters"].tolist() == [0, 0, 0, 1, 2]


def test_extract_and_translate_match_reference(oracle, golden_dir):
    z = np.load(golden_dir / "seqs.npz")
    out, off, ln = oracle.extract(z["seqs"], z["offsets"], z["ex_idx"], z["ex_starts"], z["ex_ends"], z["ex_strands"])
    assert np.array_equal(out, z["ex_seqs"]) and np.array_equal(ln, z["ex_lengths"]) and np.array_equal(
        off, z["ex_offsets"]
    )
    assert bytes(out[:6]) == b"CCGGTT" and bytes(out[6:17]) == b"acgtNAACCGG"
    for to_stop in (0, 1):
        out, off, ln = oracle.translate(z["seqs"], z["offsets"], z["lengths"], z["frames"], to_stop)
        assert np.array_equal(out, z[f"tr{to_stop}_seqs"]) and np.array_equal(ln, z[f"tr{to_stop}_lengths"])
    o = int(z["tr1_offsets"][1])  # record 1 = CATGAAANNNTTTtaaGGG read in frame 1
    assert bytes(z["tr1_seqs"][o : o + 4]) == b"MKXF" and int(z["tr1_lengths"][1]) == 4
