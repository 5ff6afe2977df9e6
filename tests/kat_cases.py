"""Hand-derived known-answer cases K16..K21: behaviours of the reference that K1..K15 do not
touch.  Expected values are worked out BY HAND from the cited lines of /root/reference/lib/src/kmergutsjava/
KmerGutsJava.java ("KGJ:n"), not produced by running anything; the derivation is in each docstring.  They are checked on
both CPU restatements (tests/test_oracle_kat.py) and on the HIP path through the C ABI (tests/test_gpu_kat.py), so a
misreading shared by the restatements cannot hide behind their agreement.  Parity stays "unpinned" (no JVM, no
reference fixture)."""
import struct

import numpy as np

A, B = 7, 9            # two function indices
HIT = np.dtype([("container", "<u4"), ("from0InProt", "<i4"), ("oI", "<i4"), ("avgOffFromEnd", "<i4"), ("fI", "<i4"),
                ("functionWt", "<f4")])
INT_MIN, INT_MAX = -(1 << 31), (1 << 31) - 1


def hits(rows):
    """rows: (from0InProt, avgOffFromEnd, fI[, oI[, functionWt]])"""
    h = np.zeros(len(rows), dtype=HIT)
    for i, r in enumerate(rows):
        h[i]["from0InProt"], h[i]["avgOffFromEnd"], h[i]["fI"] = r[0], r[1], r[2]
        h[i]["oI"] = r[3] if len(r) > 3 else 3
        h[i]["functionWt"] = r[4] if len(r) > 4 else 1.0
    return h


def k16_order_constraint_at_20_and_21():
    """-O, KGJ:490-494: a hit joins a non-empty list iff fI == last.fI and
           |(ph.from0InProt - last.from0InProt) - (last.avgOffFromEnd - ph.avgOffFromEnd)| <= 20,
    `last` = the last ACCEPTED hit (hits.get(hits.size()-1)).  All fI = A, oI = 3, weight 1, -m 5 -g 200.
    (a) d = +20 accepted, +21 rejected:
        (0, avg 100)   list empty -> accepted
        (10, 110)      d = 10 - (100 - 110) = 20          -> accepted
        (20, 120)      d = 10 - (110 - 120) = 20          -> accepted
        (30, 131)      d = 10 - (120 - 131) = 21          -> REJECTED (last stays (20, 120))
        (40, 120)      d = 20 - (120 - 120) = 20          -> accepted
        (50, 120)      d = 10                              -> accepted
        (60, 120)      d = 10                              -> accepted
      six accepted (0,10,20,40,50,60); no gap > 200, one function: only the final flush (KGJ:511-513) calls:
      CALL start 0, end 60 + 7, count 6, fI A, weight 6.0; six votes for oI 3.
    (b) d = -20 accepted, -21 rejected:
        (0, 100) accepted; (10, 70): d = 10 - (100 - 70) = -20 accepted; (20, 39): d = 10 - (70 - 39) = -21 REJECTED;
        (21, 40): d = 11 - (70 - 40) = -19 accepted; (30, 40): d = 9 accepted; (40, 40): d = 10 accepted
      five accepted (0,10,21,30,40): CALL 0, 47, 5, A, 5.0; OTU 5-3.
    (c) the same records as (a) without -O: all seven accepted: CALL 0, 67, 7, A, 7.0; OTU 7-3."""
    a = hits([(0, 100, A), (10, 110, A), (20, 120, A), (30, 131, A), (40, 120, A), (50, 120, A), (60, 120, A)])
    b = hits([(0, 100, A), (10, 70, A), (20, 39, A), (21, 40, A), (30, 40, A), (40, 40, A)])
    return [
        ("K16a", a, dict(order_constraint=True), [(0, 67, 6, A, 6.0)], [(6, 3)]),
        ("K16b", b, dict(order_constraint=True), [(0, 47, 5, A, 5.0)], [(5, 3)]),
        ("K16c", a, dict(order_constraint=False), [(0, 67, 7, A, 7.0)], [(7, 3)]),
    ]


def k17_abs_of_int_min():
    """-O, KGJ:491-494: the difference is Java int arithmetic (wraps) and Math.abs(Integer.MIN_VALUE) is
    Integer.MIN_VALUE, which is <= 20: a difference of exactly -2^31 ACCEPTS.
    (a) (0, avg 0) accepted; (10, avg 2147483638): last.avg - ph.avg = -2147483638, d = 10 + 2147483638 = 2^31 -> wraps to
        Integer.MIN_VALUE, abs = MIN_VALUE <= 20 -> accepted; (20 / 30 / 40, avg 2147483638): d = 10 -> accepted.
        Five accepted: CALL 0, 47, 5, A, 5.0; OTU 5-3.
    (b) (10, avg 2147483639) instead: d = 10 + 2147483639 = 2^31 + 1 -> wraps to -2147483647, abs = 2147483647 > 20 ->
        rejected; then (20 / 30 / 40, avg 2147483638) against last = (0, 0): d = p + 2147483638 >= 2^31 + 10 -> wraps to
        -2147483638 + (p - 20)..., |d| > 20 -> rejected.  One hit in the list at the end: no CALL, no votes."""
    big = 2147483638
    a = hits([(0, 0, A), (10, big, A), (20, big, A), (30, big, A), (40, big, A)])
    b = hits([(0, 0, A), (10, big + 1, A), (20, big, A), (30, big, A), (40, big, A)])
    return [
        ("K17a", a, dict(order_constraint=True), [(0, 47, 5, A, 5.0)], [(5, 3)]),
        ("K17b", b, dict(order_constraint=True), [], []),
    ]


def k18_cap_and_pair_rule_on_a_hit_that_was_not_appended():
    """KGJ:496-508: a hit is appended only while hits.size() < MAX_HITS_PER_SEQ - 2 = 39 998, but the pair rule right
    behind it -- hits.size() > 1 && currentFI != fI && hits[size-2].fI == hits[size-1].fI -- looks at the hit's fI and at
    the LIST's last two entries, whether or not the hit went in.
      39 998 hits of A (oI 3, weight 1) at positions 0..39997 (gaps of 1): all appended, currentFI = A.
      hit 39 998 (A, position 39998): list full -> not appended; currentFI == fI -> nothing.
      hit 39 999 (B, oI 4, position 39999): no gap (39997 + 200 >= 39999); not appended; currentFI (A) != B and the list's
        last two are A, A -> processSetOfHits(currentFI = A): 39 998 votes, float32 sum 39998.0 (exact), CALL start 0, end
        39997 + 7 = 40004; tail (KGJ:441-452): hits[n-2].fI == currentFI -> the list is CLEARED; the B hit is lost.
      hits 40 000..40 004 (B, oI 4, positions 40000..40004): list empty -> currentFI = B, five appended; final flush
        (KGJ:511-513): CALL start 40000, end 40004 + 7 = 40011, count 5, B, 5.0.
    OTU buffer: 39 998 votes for oI 3, then five for oI 4 (never ahead): [39998-3, 5-4]."""
    rows = [(p, 0, A, 3) for p in range(39998)] + [(39998, 0, A, 3), (39999, 0, B, 4)] + [(40000 + i, 0, B, 4) for i in range(5)]
    return [("K18", hits(rows), dict(), [(0, 40004, 39998, A, 39998.0), (40000, 40011, 5, B, 5.0)], [(39998, 3), (5, 4)])]


def k19_gap_test_wraps():
    """KGJ:477-478: (last.from0InProt + maxGap) < ph.from0InProt is int arithmetic.  -g 2147483647, -m 2, six hits of A at
    positions 0..5:
      pos 0 appended; pos 1: 0 + 2147483647 = 2147483647 < 1 is false -> appended (list [0, 1]);
      pos 2: 1 + 2147483647 wraps to -2147483648 < 2 -> the gap rule FIRES although the hits are adjacent: size 2 >= 2 ->
        processSetOfHits: CALL start 0, end 1 + 7 = 8, count 2, A, 2.0; cleared; pos 2 starts a new list;
      pos 3: 2 + 2147483647 wraps negative < 3 -> fires: size 1 < 2 -> cleared; likewise at 4 and 5; the final list
        holds one hit: no CALL.
    With -g 2147483646: 1 + 2147483646 = 2147483647 < 2 false; pos 3: 2 + 2147483646 wraps -> fires with the list
    [0, 1, 2]: CALL 0, 9, 3; then single-hit lists: nothing more."""
    h = hits([(p, 0, A) for p in range(6)])
    return [
        ("K19a", h, dict(max_gap=INT_MAX, min_hits=2), [(0, 8, 2, A, 2.0)], [(2, 3)]),
        ("K19b", h, dict(max_gap=INT_MAX - 1, min_hits=2), [(0, 9, 3, A, 3.0)], [(3, 3)]),
    ]


AGGREGATION_CASES = (k16_order_constraint_at_20_and_21, k17_abs_of_int_min,
                     k18_cap_and_pair_rule_on_a_hit_that_was_not_appended, k19_gap_test_wraps)


def table_image(num_sigs, entries, n_records=None):
    """kmer.table.mem_map image: header numSigs, then n_records (default numSigs) 24-byte records; entries: slot ->
    (whichKmer, oI, avgFromEnd, fI, wt); everything else empty (whichKmer = 20^8 + 1 > MAX_ENCODED, KGJ:1000)."""
    n_records = num_sigs if n_records is None else n_records
    body = bytearray()
    for i in range(n_records):
        k, o, a, f, w = entries.get(i, (20 ** 8 + 1, 0, 0, 0, 0.0))
        body += struct.pack("<qiiif", k, o, a, f, w)
    return struct.pack("<qqq", num_sigs, 24, 1) + bytes(body)


def decode(v):
    s = ""
    for _ in range(8):
        s = "ACDEFGHIKLMNPQRSTVWY"[v % 20] + s
        v //= 20
    return s


def k20_negative_key_in_a_cluster():
    """KGJ:1000-1004: a slot is empty iff whichKmer > MAX_ENCODED; a NEGATIVE whichKmer (readLongLE is signed,
    KGJ:1107-1126) is therefore occupied, matches no query (inProgress keys are >= 0) and does not stop the walk.
    numSigs 101, query k-mer 25 (home slot 25 % 101 = 25), protein "AAAAAABF" + "A" (one queried window, KGJ:912).
      table X: slot 25 = key -5, slot 26 = key 25 (oI 9, avg 4, fI 9, wt 2.5)     -> found through the negative slot
      table Y: slot 25 = key Long.MIN_VALUE, slot 26 = key 25                       -> found
      table Z: slot 25 empty, slot 26 = key 25                                      -> not found (walk ends at slot 25)"""
    q = (decode(25) + "A").encode()
    rec = (25, 9, 4, 9, 2.5)
    want = [(0, 9, 4, 9, 2.5)]                     # (from0InProt, oI, avgOffFromEnd, fI, functionWt)
    return [
        ("K20x", table_image(101, {25: (-5, 1, 1, 1, 1.0), 26: rec}), q, want),
        ("K20y", table_image(101, {25: (-(1 << 63), 1, 1, 1, 1.0), 26: rec}), q, want),
        ("K20z", table_image(101, {26: rec}), q, []),
    ]


def k21_file_longer_than_num_sigs():
    """KGJ:964-999: the table stream is read record after record; the running slot (curHashCode) is never compared with
    numSigs, so a file that holds MORE than numSigs records keeps being read past slot numSigs - 1; the stream ends at
    EOF, not at numSigs (swallowed, KGJ:799-802).  numSigs 101, query k-mer 100 (home slot 100, the last one):
      file P: 101 records, slot 100 = key 302 (also home 100: occupied, no match)  -> EOF behind it: not found
      file Q: 102 records, slot 100 = key 302, record 101 = key 100 (oI 6, avg 2, fI 5, wt 1.5) -> found in the extra record
      file R: 103 records, slot 100 = key 302, record 101 empty, record 102 = key 100   -> not found (empty slot first)"""
    q = (decode(100) + "A").encode()
    occ = (302, 1, 1, 1, 1.0)
    rec = (100, 6, 2, 5, 1.5)
    return [
        ("K21p", table_image(101, {100: occ}), q, []),
        ("K21q", table_image(101, {100: occ, 101: rec}, 102), q, [(0, 6, 2, 5, 1.5)]),
        ("K21r", table_image(101, {100: occ, 102: rec}, 103), q, []),
    ]


LOOKUP_CASES = (k20_negative_key_in_a_cluster, k21_file_longer_than_num_sigs)
