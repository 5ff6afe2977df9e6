"""Host-side mirror of the reference's public interface for the hot path.

    from kmergutsjava_amd import KmerGutsJava
    KmerGutsJava.main(["-D", data_dir, "-q", "contigs.fna", "-o", "out.txt"])      # KGJ:560-654
    KmerGutsJava().run(data_dir, fasta_path_or_None, writer, stdout)               # KGJ:742-820
    KmerGutsJava().status()                                                        # KmerGutsJavaServer.java:33-45

("KGJ:n" = reference lib/src/kmergutsjava/KmerGutsJava.java line n.)  Same flags, same data
directory layout (kmer.table.mem_map[.gz], function.index[.gz]; KGJ:749-759), same FASTA rules
(KGJ:1132-1192), same report text (KGJ:398-404, 518-548).  Everything between readFasta and the
report printers runs on the GPU through the C ABI (hotpath.py -> libkmerguts_hip.so); this module
only parses text and prints records.  There is no CPU fallback: without the library or without a
GPU, run() raises.

Known, documented deviations from the reference's behaviour:
  * the progress lines "Processed: NN%, time=..." of the table stream (KGJ:1019-1025) are not
    printed: the table is not streamed;
  * -d: the HIT / after-hit / after-call stream is printed (from the library's event bytes), and so are
    the info lines incl. "Kmers found: N (pos-count=M)" (KGJ:1031-1033: N = distinct matched k-mer values,
    M = hit records, over every FASTA record incl. records shadowed by a later one of the same id); only
    the time-stamped progress lines of the reference's table stream are not (KGJ:1019-1025); when a query walks
    off the end of the table the reference's stream throws EOFException and run() prints "Error: null" instead of
    "Kmers found" (KGJ:797-802): so does this mirror (kg_stats.lookup_ran_off); a truncated table file makes a
    .gz stream fail with "Error skipping N bytes" instead -- here it is "Error: null" as well;
  * "Temp. directory:" prints the canonical path of /tmp, the JVM's java.io.tmpdir on Linux whatever
    $TMPDIR says (KGJ:108, 744-748; -t can never change it, see below);
  * -t / -l: the reference's switch falls through to "Unknown parameter" for both (KGJ:605-611);
    this mirror does the same (message + usage, then carries on, KGJ:616-647);
  * input beyond 20 M k-mers: the reference silently drops queries in its external merge
    (KGJ:705-709); here nothing is dropped.
"""
from __future__ import annotations

import gzip
import io
import os
import sys
import time
from decimal import Decimal, ROUND_HALF_UP
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from . import _native as N
from .hotpath import Params, SignatureTable

_WS = " \t\n\x0b\x0c\r\x1c\x1d\x1e\x1f"      # what String.trim() strips is every char <= ' ' ; see _trim


def _trim(s: str) -> str:
    a, b = 0, len(s)
    while a < b and s[a] <= " ":
        a += 1
    while b > a and s[b - 1] <= " ":
        b -= 1
    return s[a:b]


def _read_lines(text: str) -> List[str]:
    """BufferedReader.readLine(): \\n, \\r or \\r\\n end a line; no empty line after a final terminator."""
    if not text:
        return []
    lines = text.replace("\r\n", "\n").replace("\r", "\n").split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    return lines


def _parse_int(x: Optional[str]) -> int:
    """Integer.parseInt with the reference's failure text."""
    try:
        if x is None:
            raise ValueError
        return int(x, 10)
    except ValueError:
        raise ValueError('For input string: "%s"' % x if x is not None else "null")


def read_fasta(text: str, callback: Callable[[str, str, str], None]) -> None:
    """readFasta (KGJ:1132-1192): id = first token after '>' (split on space / tab), sequence lines
    concatenated untrimmed, blank lines before the sequence skipped, malformed input raises."""
    lines = _read_lines(text)
    n = len(lines)
    i = 0
    pending: Optional[str] = None          # str1 carried over from the previous record
    have_pending = False
    while True:
        name = None
        descr = ""
        if not have_pending:
            pending = lines[i] if i < n else None
            i += 1
        have_pending = False
        while pending is not None:
            t = _trim(pending)
            if len(t) > 1:
                if t[0] == ">" and len(_trim(t[1:])) > 0:
                    toks = [x for x in t[1:].replace("\t", " ").split(" ") if x]
                    name, descr = toks[0], " ".join(toks[1:])
                    break
                raise ValueError("Wrong caption line: " + t)
            pending = lines[i] if i < n else None
            i += 1
        if name is None:
            return
        while True:
            pending = lines[i] if i < n else None
            i += 1
            if pending is None or _trim(pending).startswith(">"):
                raise ValueError("No sequence for caption: " + name)
            if len(_trim(pending)) > 0:
                break
        parts = []
        while True:
            parts.append(pending)
            pending = lines[i] if i < n else None
            i += 1
            if pending is None or _trim(pending).startswith(">"):
                break
        have_pending = True
        seq = "".join(parts)
        if not seq:
            raise ValueError("No sequence for caption: " + name)
        callback(name, seq, descr)


def load_indexed_array(text: str) -> List[str]:
    """loadIndexedArray (KGJ:345-369): '<i>\\t<value>' lines, dense and in order."""
    out = []
    for pos, line in enumerate(_read_lines(text)):
        tab = line.find("\t")
        if tab < 0:
            raise IndexError("String index out of range: -1")      # substring(0, -1) in the reference
        if int(line[:tab]) != pos:
            raise ValueError("Your index must be dense and in order (see line %d)" % pos)
        out.append(line[tab + 1:])
    return out


def java_format_f(v, precision: int = 6) -> str:
    """String.format("%f") of a float: the decimal digits of (double)v rounded HALF_UP
    (java.util.Formatter; differs from C's round-half-even on exact ties such as 5.0078125f)."""
    v = float(np.float32(v))
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "Infinity" if v > 0 else "-Infinity"
    d = Decimal(v).quantize(Decimal(1).scaleb(-precision), rounding=ROUND_HALF_UP)
    s = format(d, "f")
    if d == 0 and np.signbit(v) and not s.startswith("-"):
        s = "-" + s
    return s


def _read_text(path: str) -> str:
    if path.endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return f.read().decode("latin-1")
    with open(path, "rb") as f:
        return f.read().decode("latin-1")


_AA_LUT = np.full(256, 20, dtype=np.int64)
for _i, _c in enumerate("ACDEFGHIKLMNPQRSTVWY"):
    _AA_LUT[ord(_c)] = _i
_BASE_LUT = np.full(256, 4, dtype=np.int64)
for _c, _v in (("a", 0), ("A", 0), ("c", 1), ("C", 1), ("g", 2), ("G", 2), ("t", 3), ("T", 3), ("u", 3), ("U", 3)):
    _BASE_LUT[ord(_c)] = _v
_CODON_AA = np.array([_AA_LUT[ord(c)] for c in "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF"], dtype=np.int64)
_POW20 = 20 ** np.arange(7, -1, -1, dtype=np.int64)


def hit_kmer_values(seq: bytes, aa: bool, positions_per_container) -> np.ndarray:
    """encodedKmer (KGJ:274-292) of the window behind every hit record of one sequence: positions_per_container =
    from0InProt arrays of its 1 (protein) or 6 (+0 +1 +2 -0 -1 -2, KGJ:1064-1072) containers.  Host side, for the
    -d line "Kmers found" only (the hit records do not carry the k-mer)."""
    s = np.frombuffer(seq, dtype=np.uint8)
    out = []
    if aa:
        p = np.asarray(positions_per_container[0], dtype=np.int64)
        out.append((_AA_LUT[s[p[:, None] + np.arange(8)[None, :]]] * _POW20).sum(axis=1))
    else:
        L = len(s)
        b = _BASE_LUT[s]
        for f in range(6):
            p = np.asarray(positions_per_container[f], dtype=np.int64)
            start = (f % 3) + 3 * p                                        # first base of the window on its strand
            k = start[:, None] + np.arange(24)[None, :]
            codes = b[k] if f < 3 else 3 - b[L - 1 - k]                    # revComp (KGJ:263-272): compl code = 3 - code
            c = codes.reshape(-1, 8, 3)
            out.append((_CODON_AA[c[:, :, 0] * 16 + c[:, :, 1] * 4 + c[:, :, 2]] * _POW20).sum(axis=1))
    return np.concatenate(out) if out else np.zeros(0, np.int64)


_TABLES: Dict[Tuple[str, float, int, int], SignatureTable] = {}     # tables stay resident in HBM across run() calls


def _resident_table(path: str, device: int) -> SignatureTable:
    st = os.stat(path)
    key = (os.path.realpath(path), st.st_mtime, st.st_size, device)
    tab = _TABLES.get(key)
    if tab is None:
        tab = SignatureTable.open(path, device)        # plain or .gz (KGJ:749-753): the library streams it to the device
        _TABLES[key] = tab
    return tab


class Hit:
    """KGJ:1213-1219"""
    __slots__ = ("oI", "from0InProt", "avgOffFromEnd", "fI", "functionWt")

    def __init__(self, oI=0, from0InProt=0, avgOffFromEnd=0, fI=0, functionWt=0.0):
        self.oI, self.from0InProt, self.avgOffFromEnd, self.fI, self.functionWt = oI, from0InProt, avgOffFromEnd, fI, functionWt


class OtuCount:
    """KGJ:1221-1224"""
    __slots__ = ("oI", "count")

    def __init__(self, oI=0, count=0):
        self.oI, self.count = oI, count


class HitContainerKey:
    """KGJ:1226-1260: equality and hash over (queryId, strand, frame)."""
    __slots__ = ("queryId", "strand", "frame")

    def __init__(self, queryId=None, strand="+", frame=0):
        self.queryId, self.strand, self.frame = queryId, strand, frame

    def __eq__(self, other):
        return (isinstance(other, HitContainerKey) and self.frame == other.frame and self.queryId == other.queryId
                and self.strand == other.strand)

    def __hash__(self):
        return hash((self.queryId, self.strand, self.frame))


class HitContainer:
    """KGJ:1262-1266"""
    __slots__ = ("key", "id", "hits")

    def __init__(self, key=None, id=0, hits=None):
        self.key, self.id, self.hits = key, id, [] if hits is None else hits


class QueryKmer:
    """KGJ:1200-1204"""
    __slots__ = ("value", "hitCntId", "protPos")

    def __init__(self, value=0, hitCntId=0, protPos=0):
        self.value, self.hitCntId, self.protPos = value, hitCntId, protPos


def _hits_to_records(hits, container: int = 0) -> np.ndarray:
    rec = np.zeros(len(hits), dtype=N.HIT_DTYPE)
    for i, h in enumerate(hits):
        rec[i] = (container, h.from0InProt, h.oI, h.avgOffFromEnd, h.fI, h.functionWt)
    return rec


def _otu_to_record(oICounts) -> np.ndarray:
    rec = np.zeros(1, dtype=N.OTU_DTYPE)
    if len(oICounts) > KmerGutsJava.OI_BUFSZ:
        raise ValueError("oICounts holds more than OI_BUFSZ entries")
    rec["n"][0] = len(oICounts)
    for j, oc in enumerate(oICounts):
        rec["oI"][0][j], rec["count"][0][j] = oc.oI, oc.count
    return rec


def _record_to_otu(rec, oICounts) -> None:
    oICounts[:] = [OtuCount(int(rec["oI"][j]), int(rec["count"][j])) for j in range(int(rec["n"]))]


class KmerGutsJava:
    Hit, OtuCount, HitContainerKey, HitContainer, QueryKmer = Hit, OtuCount, HitContainerKey, HitContainer, QueryKmer

    # KGJ:85-99
    K = 8
    CORE = 20 ** 7
    MAX_ENCODED = 20 ** 8
    GENETIC_CODE = tuple("KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF")
    PROT_ALPHA = tuple("ACDEFGHIKLMNPQRSTVWY")
    VERSION = 1
    MAX_HITS_PER_SEQ = 40000
    OI_BUFSZ = 5

    # one batch handed to the GPU: below the C ABI's 2^32-256 windows per call
    MAX_BATCH_CHARS = 1_500_000_000

    def __init__(self, device: int = 0):
        # KGJ:102-109
        self.aa = False
        self.orderConstraint = False
        self.minHits = 5
        self.minWeightedHits = 0
        self.maxGap = 200
        self.debug = False
        self.device = device
        self.last_stats: List[dict] = []

    # ---- the static helpers a caller of the reference class could use (KGJ:111-318) ----
    @staticmethod
    def toAminoAcidOff(c: str) -> int:
        i = "ACDEFGHIKLMNPQRSTVWY".find(c) if len(c) == 1 else -1
        return i if i >= 0 else 20

    _COMPL = {"a": "t", "A": "T", "c": "g", "C": "G", "g": "c", "G": "C", "t": "a", "u": "a", "T": "A", "U": "A",
              "m": "k", "M": "K", "r": "y", "R": "Y", "w": "w", "W": "W", "s": "S", "S": "S", "y": "r", "Y": "R",
              "k": "m", "K": "M", "b": "v", "B": "V", "d": "h", "D": "H", "h": "d", "H": "D", "v": "b", "V": "B",
              "n": "n", "N": "N"}

    @staticmethod
    def compl(c: str) -> str:
        return KmerGutsJava._COMPL.get(c, c)

    @staticmethod
    def revComp(data) -> str:
        return "".join(KmerGutsJava._COMPL.get(c, c) for c in reversed("".join(data)))

    @staticmethod
    def encodedKmer(data, pos: int) -> int:
        enc = 0
        for i in range(8):
            add = data[pos + i]
            if add >= 20:
                return -1
            enc = enc * 20 + add
        return enc

    @staticmethod
    def dnaChar(c: str) -> int:
        return {"a": 0, "A": 0, "c": 1, "C": 1, "g": 2, "G": 2, "t": 3, "u": 3, "T": 3, "U": 3}.get(c, 4)

    # ---- the public instance methods of the reference class (KGJ:385, 457, 526, 1082), on the GPU through the C ABI ----
    def _params(self) -> Params:
        return Params(aa=True, order_constraint=self.orderConstraint, min_hits=self.minHits,
                      min_weighted_hits=self.minWeightedHits, max_gap=self.maxGap)

    def processSetOfHits(self, hits: list, functionArray, currentFI: int, oICounts: list, pw) -> int:
        """KGJ:385-455: one step of the state machine on the caller's list; prints the CALL (and, with debug, the
        after-call line), updates oICounts, keeps the last two hits or empties the list, returns the new currentFI."""
        import ctypes as C
        if len(hits) < 2:
            raise IndexError("Index: %d, Size: %d" % (len(hits) - 2, len(hits)))      # hits.get(numHits - 2) throws
        rec = _hits_to_records(hits)
        otu = _otu_to_record(oICounts)
        call = np.zeros(1, dtype=N.CALL_DTYPE)
        called, new_fi, keep2 = C.c_int32(), C.c_int32(), C.c_int32()
        p = self._params().to_native()
        N.check(N.load().kg_process_set_of_hits(self.device, C.byref(p), rec.ctypes.data, len(rec), int(currentFI),
                                                 otu.ctypes.data, call.ctypes.data, C.byref(called), C.byref(new_fi), C.byref(keep2)))
        if called.value:
            self._print_call(call[0], functionArray, pw.write)
            if self.debug:                                                          # KGJ:406-409
                pw.write("after-call: hits: " + "".join("%d/%s/%d " % (h.from0InProt, java_format_f(h.functionWt), h.fI) for h in hits) + "\n")
        _record_to_otu(otu[0], oICounts)
        hits[:] = hits[-2:] if keep2.value else []
        return int(new_fi.value)

    def gatherHits(self, ln_DNA: int, strand: str, frame: int, allHits: list, functionArray, oICounts: list, pw) -> None:
        """KGJ:457-514: sorts allHits by from0InProt (in place, stable), runs the run / vote state machine over them,
        prints the CALL lines (the whole -d stream with debug) and carries the OTU buffer oICounts on."""
        from .hotpath import aggregate_hits
        allHits.sort(key=lambda h: h.from0InProt)                                   # Collections.sort is stable (KGJ:460-465)
        rec = _hits_to_records(allHits)
        with aggregate_hits(rec, [0, len(rec)], 1, self._params(), _otu_to_record(oICounts), self.device) as r:
            calls, otu = r.calls(), r.otu()
            if self.debug:
                self._print_debug_stream(calls, functionArray, pw.write, rec, r.hit_events(), int(r.container_tail_events()[0]))
            else:
                self._print_calls(calls, functionArray, pw.write)
        _record_to_otu(otu[0], oICounts)

    def processAASeq(self, id: str, proteinLen: int, hitCnts: dict, functionArray, pw) -> None:
        """KGJ:526-536"""
        oICounts: list = []
        pw.write("PROTEIN-ID\t%s\t%d\n" % (id, proteinLen))
        self.gatherHits(proteinLen, "+", 0, hitCnts[HitContainerKey(id, "+", 0)].hits, functionArray, oICounts, pw)
        pw.write("OTU-COUNTS\t%s[%d]" % (id, proteinLen) + "".join("\t%d-%d" % (oc.count, oc.oI) for oc in oICounts) + "\n")   # KGJ:516-524

    @staticmethod
    def createKmerComparator(numSigs: int):
        """KGJ:1082-1095: orders QueryKmer objects by (value % numSigs, value); a cmp function (functools.cmp_to_key)."""
        def compare(o1, o2) -> int:
            h1, h2 = o1.value % numSigs, o2.value % numSigs
            if h1 != h2:
                return -1 if h1 < h2 else 1
            return (o1.value > o2.value) - (o1.value < o2.value)
        return compare

    # ---- KmerGutsJavaServer.status (KmerGutsJavaServer.java:33-45) ----
    def status(self) -> dict:
        return {"state": "OK", "message": "", "version": "0.0.1", "git_url": "", "git_commit_hash": ""}

    # ---- printInfoLine (KGJ:891-898) ----
    def _info(self, message: str, pw, stdout: bool) -> None:
        if self.debug:
            pw.write(message + "\n")
        if not stdout:
            print(message)

    # ---- run (KGJ:742-820) ----
    def run(self, kmerTableDir: str, queryFastaFile: Optional[str], pw, stdout: bool) -> None:
        self._info("Temp. directory: " + os.path.realpath("/tmp"), pw, stdout)     # KGJ:108: java.io.tmpdir
        table_path = os.path.join(kmerTableDir, "kmer.table.mem_map")
        if os.path.exists(table_path + ".gz"):
            table_path += ".gz"                                   # KGJ:750-753: the .gz wins
        fn_path = os.path.join(kmerTableDir, "function.index")
        if os.path.exists(fn_path + ".gz"):
            fn_path += ".gz"
        function_array = load_indexed_array(_read_text(fn_path))
        text = sys.stdin.read() if queryFastaFile is None else _read_text(queryFastaFile)
        tab = _resident_table(table_path, self.device)

        t1 = time.time()
        ids: List[str] = []
        seqs: List[bytes] = []
        read_fasta(text, lambda name, seq, descr: (ids.append(name), seqs.append(seq.encode("latin-1"))))
        self._info("Preparation time: %d ms." % int((time.time() - t1) * 1000), pw, stdout)

        # queryIdToLen / hitCnts are maps (KGJ:772, 805-809): a repeated id is reported once, at the place
        # of its first record, with the length and the hits of its last record
        last_of: Dict[str, int] = {}
        for k, name in enumerate(ids):
            last_of[name] = k
        order = [last_of[name] for name in last_of]

        t2 = time.time()
        if self.debug:                                            # KGJ:951-954
            pw.write("Kmer-table info: numSigs=%(numSigs)d, entrySize=%(entrySize)d, version=%(version)d\n" % tab.info())
        # the info lines of the lookup ("Processed: NN%", "Kmers found", "Error: ...") are printed when -d is given or the
        # report goes to a file (printInfoLine, KGJ:891-898); they need what the reference's table stream would have seen
        # (KG_F_PROGRESS), over every FASTA record.  Otherwise the plain, faster scan.
        need_info = self.debug or not stdout
        params = Params(aa=self.aa, order_constraint=self.orderConstraint, min_hits=self.minHits,
                        min_weighted_hits=self.minWeightedHits, max_gap=self.maxGap, progress=need_info)
        per = 1 if self.aa else 6
        results = {}                    # record index -> (calls of its containers, otu record)
        self.last_stats = []
        batch: List[int] = []
        size = 0
        progress: List[dict] = []               # per batch: kg_progress
        found_slots: List[np.ndarray] = []      # per batch: the distinct table slots of its hit records
        pos_count = 0                           # hit records (KGJ:1014)

        def flush():
            nonlocal batch, size, pos_count
            if not batch:
                return
            off = np.zeros(len(batch) + 1, dtype=np.int64)
            np.cumsum([len(seqs[k]) for k in batch], out=off[1:])
            buf = np.frombuffer(b"".join(seqs[k] for k in batch), dtype=np.uint8)
            with tab.scan(buf, off, params) as r:
                calls, ccs, otu = r.calls(), r.container_call_start(), r.otu()
                if self.debug:      # -d: the hit records and what gatherHits did at each of them
                    hits, chs, ev, tail = r.hits(), r.container_hit_start(), r.hit_events(), r.container_tail_events()
                self.last_stats.append(r.stats)
                if need_info:
                    progress.append(r.progress())
                    pos_count += r.stats["n_hits"]
                    found_slots.append(np.unique(r.hit_slots()) if size_total > self.MAX_BATCH_CHARS else None)
            for j, k in enumerate(batch):
                cs = range(j * per, j * per + per)
                dbg = [(hits[chs[c]:chs[c + 1]], ev[chs[c]:chs[c + 1]], int(tail[c])) for c in cs] if self.debug else None
                results[k] = ([calls[ccs[c]:ccs[c + 1]] for c in cs], otu[j], dbg)
            batch, size = [], 0

        # -d counts what the reference's lookup counts: every FASTA record has containers of its own there, also a
        # record that a later one of the same id shadows in the report (KGJ:805-809)
        scan_order = range(len(ids)) if need_info else order
        size_total = sum(len(seqs[k]) for k in scan_order)        # (more than one batch: the distinct slots are combined here)
        for k in scan_order:
            if batch and size + len(seqs[k]) > self.MAX_BATCH_CHARS:
                flush()
            batch.append(k)
            size += len(seqs[k])
        flush()
        if need_info:
            self._lookup_info(progress, found_slots, pos_count, table_path.endswith(".gz"), t2, pw, stdout)
        self._info("Lookup time: %d ms." % int((time.time() - t2) * 1000), pw, stdout)

        t3 = time.time()
        for k in order:
            calls, otu, dbg = results[k]
            self.write_record(pw, ids[k], len(seqs[k]), calls, otu, function_array, dbg)
        pw.flush()
        self._info("Grouping time: %d ms." % int((time.time() - t3) * 1000), pw, stdout)

    def _lookup_info(self, progress: List[dict], found_slots, pos_count: int, gz: bool, t2: float, pw, stdout: bool) -> None:
        """What lookup prints besides the records (KGJ:1016-1033) and how run() reports its failure (KGJ:797-802), from the
        scans' kg_progress: the merge-join visits the table in slot order, so the batches' summaries combine by minimum and
        maximum; one "Processed" line per tenth of the table in which a slot was visited, at the first such slot."""
        if not progress:
            return
        first = [min([p["first_visited"][f] for p in progress if p["first_visited"][f] >= 0], default=-1) for f in range(11)]
        last = max(p["last_visited"] for p in progress)
        beyond = min([p["first_beyond"] for p in progress if p["first_beyond"] >= 0], default=-1)
        ran_off = any(p["walk_ran_off"] for p in progress)
        stream_slots = progress[0]["stream_slots"]
        if len(progress) == 1:
            found_upto, kmers_found = progress[0]["found_upto"], progress[0]["kmers_found"]
        else:                                                     # a k-mer found in two batches counts once
            slots = np.unique(np.concatenate([x for x in found_slots if x is not None])) if any(x is not None for x in found_slots) else np.zeros(0, np.uint32)
            found_upto = [int(np.searchsorted(slots, first[f], side="right")) if first[f] >= 0 else 0 for f in range(11)]
            kmers_found = len(slots)
        for f in range(1, 11):                                    # (tenth 0 is where the join starts: never a change)
            if first[f] >= 0:
                self._info("Processed: %d%%, time=%d ms., found-so-far=%d" % (f * 10, int((time.time() - t2) * 1000), found_upto[f]),
                           pw, stdout)
        if ran_off:
            # a query walked to the end of the table undecided: there the reference's stream throws EOFException,
            # which run() reports and swallows (KGJ:797-802); the hits are the same, "Kmers found" is not reached
            self._info("Error: null", pw, stdout)
        elif beyond >= 0:
            # the table stream is shorter than numSigs records and a query's home slot lies behind its end: the join skips
            # to it -- a GZIPInputStream comes up short ("Error skipping N bytes", KGJ:1036-1049), a plain file seeks past
            # its end and the read behind it throws EOFException
            # (a home slot right AT the end of the stream is skipped to without trouble; the read there is what fails)
            skip = 24 * (beyond - (last + 1))
            self._info("Error: Error skipping %d bytes" % skip if gz and beyond > stream_slots else "Error: null", pw, stdout)
        elif self.debug:                                          # KGJ:1031-1033
            pw.write("Kmers found: %d (pos-count=%d)\n" % (kmers_found, pos_count))

    def write_record(self, pw, name: str, ln: int, calls_per_container, otu, function_array, debug_per_container=None) -> None:
        """The report of one sequence: processSeq / processAASeq + tabulateOtuDataForContig
        (KGJ:526-558, 516-524).  calls_per_container: 1 (protein) or 6 (+0 +1 +2 -0 -1 -2) CALL record arrays;
        debug_per_container (with -d): (hit records, event bytes, tail event) per container."""
        w = pw.write

        def container(f):
            if debug_per_container is None:
                self._print_calls(calls_per_container[f], function_array, w)
            else:
                self._print_debug_stream(calls_per_container[f], function_array, w, *debug_per_container[f])

        if self.aa:
            w("PROTEIN-ID\t%s\t%d\n" % (name, ln))                              # KGJ:529
            container(0)
        else:
            w("processing %s[%d]\n" % (name, ln))                               # KGJ:541
            for f in range(6):
                w("TRANSLATION\t%s\t%d\t%s\t%d\n" % (name, ln, "+-"[f // 3], f % 3))   # KGJ:545
                container(f)
        w("OTU-COUNTS\t%s[%d]" % (name, ln))                                    # KGJ:518-522
        for j in range(int(otu["n"])):
            w("\t%d-%d" % (int(otu["count"][j]), int(otu["oI"][j])))
        w("\n")

    @staticmethod
    def _print_call(c, function_array, w) -> None:                                  # KGJ:398-404
        fi = int(c["fI"])
        if fi < 0 or fi >= len(function_array):
            raise IndexError("Index: %d, Size: %d" % (fi, len(function_array)))      # functionArray.get() throws
        w("CALL\t%d\t%d\t%d\t%d\t%s\t%s\n" % (int(c["start"]), int(c["end"]), int(c["count"]), fi,
                                             function_array[fi], java_format_f(c["weightedHits"])))

    @staticmethod
    def _print_calls(calls, function_array, w) -> None:
        for c in calls:
            KmerGutsJava._print_call(c, function_array, w)

    @staticmethod
    def _print_debug_stream(calls, function_array, w, hits, events, tail) -> None:
        """The -d text of one container (KGJ:470-473 HIT, 498-501 after-hit, 406-409 after-call, displayHits
        KGJ:376-383).  Nothing is decided here: the event bytes written by the aggregation kernel say when the
        hits list grew, was processed, kept its last two members or was emptied; this only prints."""
        live: List[int] = []                  # indices (into hits) of the reference's "hits" list
        nxt = 0

        def show(tag):
            w(tag + "hits: ")
            for i in live:
                h = hits[i]
                w("%d/%s/%d " % (int(h["from0InProt"]), java_format_f(h["functionWt"]), int(h["fI"])))
            w("\n")

        def reset(called, keep2):
            nonlocal nxt, live
            if called:
                KmerGutsJava._print_call(calls[nxt], function_array, w)
                nxt += 1
                show("after-call: ")
            live = live[-2:] if keep2 else []

        for i in range(len(hits)):
            h, e = hits[i], int(events[i])
            w("HIT\t%d\t%d\t%d\t%d\t%s\t%d\n" % (int(h["from0InProt"]), 0, int(h["avgOffFromEnd"]), int(h["fI"]),
                                                  java_format_f(h["functionWt"], 3), int(h["oI"])))
            if e & N.EV_RESET_BEFORE:
                reset(e & N.EV_CALL_BEFORE, e & N.EV_KEEP2_BEFORE)
            if e & N.EV_ACCEPTED:
                live.append(i)
                show("after-hit: ")
            if e & N.EV_RESET_AFTER:
                reset(e & N.EV_CALL_AFTER, e & N.EV_KEEP2_AFTER)
        if tail & N.EV_TAIL_CALL:
            reset(True, False)
        assert nxt == len(calls), "event bytes and CALL records disagree"

    # ---- main (KGJ:560-654) ----
    USAGE = (
        "Usage: kmer_guts [options] -D DataDir",
        "Arguments:",
        " -a - (optional) amino acids in input FASTA (default is DNA)",
        " -d - (optional) print debug messages",
        " -m - (optional) min. number of hits in result (integer, default = 5)",
        " -M - (optional) min. sum of hit weights (integer, default = 0)",
        " -O - (optional) order constraint (don't use order by default)",
        " -g - (optional) max. gap between hits to be joined (integer, default = 200)",
        " -D - (required) data directory with kmer-table and function-index files",
        " -q - (optional) query fasta file (STDIN if not defined)",
        " -o - (optional) output file (STDOUT if not defined)",
        " -t - (optional) temporary directory (system one is used by default)",
        " -l - (optional) limit for input Kmer array (long, default = 20,000,000)",
    )

    @classmethod
    def main(cls, args: List[str], device: int = 0) -> None:
        kmer_table_dir = query = output = None
        inst = cls(device)
        try:
            params = list(args)
            while params:
                param = params.pop(0)
                if not param.startswith("-"):
                    raise ValueError("Parameter name should start from '-': " + param)
                param = param[1:]
                if len(param) != 1:
                    raise ValueError("Unknown parameter: -" + param)
                ch = param
                if ch == "a":
                    inst.aa = True
                elif ch == "d":
                    inst.debug = True
                elif ch == "m":
                    inst.minHits = _parse_int(params.pop(0) if params else None)
                elif ch == "M":
                    inst.minWeightedHits = _parse_int(params.pop(0) if params else None)
                elif ch == "O":
                    inst.orderConstraint = True
                elif ch == "g":
                    inst.maxGap = _parse_int(params.pop(0) if params else None)
                elif ch == "D":
                    kmer_table_dir = params.pop(0) if params else None
                elif ch == "q":
                    query = params.pop(0) if params else None
                elif ch == "o":
                    output = params.pop(0) if params else None
                elif ch in "tl":
                    # KGJ:605-611: both cases consume their value(s) and fall through to the default branch
                    if ch == "t" and params:
                        params.pop(0)
                    _parse_int(params.pop(0) if params else None)          # Long.parseLong(params.poll())
                    raise ValueError("Unknown parameter: -" + param)
                else:
                    raise ValueError("Unknown parameter: -" + param)
            if kmer_table_dir is None:
                raise ValueError("-D parameter is required")
        except Exception as ex:                                   # KGJ:616-636: print usage, then carry on
            print("Error: " + str(ex))
            for line in cls.USAGE:
                print(line)
        if kmer_table_dir is None:
            raise TypeError("kmerTableDir is null")               # new File((String) null) throws in the reference
        if query is None:
            raise TypeError("queryFastaFile is null")             # KGJ:647: new File(null) -> NullPointerException
        if output is not None:
            with open(output, "w", newline="") as pw:
                inst.run(kmer_table_dir, query, pw, False)
        else:
            inst.run(kmer_table_dir, query, sys.stdout, True)
            sys.stdout.flush()


def main(argv: Optional[List[str]] = None) -> None:
    KmerGutsJava.main(sys.argv[1:] if argv is None else argv)


if __name__ == "__main__":
    main()
