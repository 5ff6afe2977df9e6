"""kgj_model.py -- TEST INFRASTRUCTURE ONLY.

Second, independently written restatement of the reference hot path, in plain Python,
statement by statement ("KGJ:n" = /root/reference/lib/src/kmergutsjava/KmerGutsJava.java:n).
It exists to cross-check oracle/kg_oracle.c on small inputs and to produce the Java-exact
report text (R16) that the host wrapper must reproduce.  It is slow on purpose: no numpy,
no shortcuts, the literal sorted merge-join lookup (KGJ:944-1034).

PARITY STATUS: parity unpinned -- the reference has no golden vectors for this path and
no JVM exists in the build image.  Pinned only by hand-derived KATs.

Only tests/ may import this module.
"""
from __future__ import annotations

import io
import struct
from decimal import Decimal, ROUND_HALF_UP
from functools import cmp_to_key

# KGJ:85-99
K = 8
CORE = 20 ** 7
MAX_ENCODED = CORE * 20
GENETIC_CODE = (
    "KNKNTTTTRSRSIIMI"
    "QHQHPPPPRRRRLLLL"
    "EDEDAAAAGGGGVVVV"
    "*Y*YSSSS*CWCLFLF"
)
PROT_ALPHA = "ACDEFGHIKLMNPQRSTVWY"
MAX_HITS_PER_SEQ = 40000
OI_BUFSZ = 5


def _i32(x: int) -> int:
    """Java int wrap-around."""
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def _f32(x: float) -> float:
    """Round a Python float to IEEE float32 (Java float arithmetic)."""
    return struct.unpack("<f", struct.pack("<f", x))[0]


def to_amino_acid_off(c: str) -> int:
    """KGJ:111-175"""
    i = PROT_ALPHA.find(c) if len(c) == 1 else -1
    return i if i >= 0 else 20


_COMPL = {  # KGJ:177-260, including 's' -> 'S' (KGJ:218-219)
    "a": "t", "A": "T", "c": "g", "C": "G", "g": "c", "G": "C",
    "t": "a", "u": "a", "T": "A", "U": "A",
    "m": "k", "M": "K", "r": "y", "R": "Y", "w": "w", "W": "W",
    "s": "S", "S": "S", "y": "r", "Y": "R", "k": "m", "K": "M",
    "b": "v", "B": "V", "d": "h", "D": "H", "h": "d", "H": "D",
    "v": "b", "V": "B", "n": "n", "N": "N",
}


def compl(c: str) -> str:
    return _COMPL.get(c, c)


def rev_comp(data: str) -> str:
    """KGJ:263-272"""
    return "".join(compl(c) for c in reversed(data))


def encoded_kmer(data, pos: int) -> int:
    """KGJ:274-292"""
    enc = 0
    for i in range(K):
        add = data[pos + i]
        if add >= 20:
            return -1
        enc = enc * 20 + add
    if enc > MAX_ENCODED:
        raise RuntimeError("bad encoding")
    return enc


def dna_char(c: str) -> int:
    """KGJ:294-318"""
    if c in "aA":
        return 0
    if c in "cC":
        return 1
    if c in "gG":
        return 2
    if c in "tuTU":
        return 3
    return 4


def translate(seq: str, off: int, pseq: list, p_iseq: list) -> None:
    """KGJ:320-343 (buffers are caller-owned and reused)."""
    mx = len(seq) - 3
    p = 0
    i = off
    while i <= mx:
        c1 = dna_char(seq[i]); c2 = dna_char(seq[i + 1]); c3 = dna_char(seq[i + 2])
        i += 3
        if c1 < 4 and c2 < 4 and c3 < 4:
            prot_c = GENETIC_CODE[c1 * 16 + c2 * 4 + c3]
            pseq[p] = prot_c
            p_iseq[p] = to_amino_acid_off(prot_c)
        else:
            pseq[p] = "x"
            p_iseq[p] = 20
        p += 1
    if p < len(pseq):
        pseq[p] = "\0"
        p_iseq[p] = 21


class Hit:  # KGJ:1213-1219
    __slots__ = ("oI", "from0InProt", "avgOffFromEnd", "fI", "functionWt")

    def __init__(self, oI=0, from0InProt=0, avgOffFromEnd=0, fI=0, functionWt=0.0):
        self.oI = oI; self.from0InProt = from0InProt; self.avgOffFromEnd = avgOffFromEnd
        self.fI = fI; self.functionWt = functionWt


def java_format_f(v: float, precision: int = 6) -> str:
    """String.format("%f") of a float promoted to double: HALF_UP on the decimal expansion
    (SURVEY 8c N3)."""
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "Infinity" if v > 0 else "-Infinity"
    q = Decimal(1).scaleb(-precision)
    d = Decimal(v).quantize(q, rounding=ROUND_HALF_UP)
    s = format(d, "f")
    if d == 0 and struct.pack(">d", v)[0] & 0x80 and not s.startswith("-"):
        s = "-" + s
    return s


class SkipError(Exception):
    """IllegalStateException("Error skipping N bytes") of skipBytesFully, KGJ:1045-1047."""


class Model:
    """Instance fields of KGJ:102-109 + the methods that read them.  gz: the table is read through a GZIPInputStream
    (kmer.table.mem_map.gz, KGJ:927-929), which matters only for how a stream shorter than numSigs records fails."""

    def __init__(self, aa=False, order_constraint=False, min_hits=5, min_weighted_hits=0,
                 max_gap=200, debug=False, gz=False):
        self.gz = gz
        self.aa = aa
        self.orderConstraint = order_constraint
        self.minHits = min_hits
        self.minWeightedHits = min_weighted_hits
        self.maxGap = max_gap
        self.debug = debug
        # records collected besides the text
        self.calls = []   # (container, start, end, count, fI, weightedHits)
        self.otus = []    # per sequence: [(count, oI), ...]
        self.hits = []    # (container, from0InProt, oI, avgOffFromEnd, fI, functionWt)
        self._container = 0

    # ---- KGJ:375-383
    def display_hits(self, hits, pw):
        pw.write("hits: ")
        for h in hits:
            pw.write("%d/%s/%d " % (h.from0InProt, java_format_f(h.functionWt), h.fI))
        pw.write("\n")

    # ---- KGJ:385-455
    def process_set_of_hits(self, hits, function_array, current_fi, oi_counts, pw):
        fi_count = 0
        weighted = 0.0
        last_hit = 0
        for i in range(len(hits)):
            if hits[i].fI == current_fi:
                last_hit = i
                fi_count += 1
                weighted = _f32(weighted + hits[i].functionWt)
        if fi_count >= self.minHits and weighted >= float(self.minWeightedHits):
            pw.write("CALL\t%d\t%d\t%d\t%d\t%s\t%s\n" % (
                hits[0].from0InProt, hits[last_hit].from0InProt + (K - 1), fi_count, current_fi,
                function_array[current_fi], java_format_f(weighted)))
            self.calls.append((self._container, hits[0].from0InProt, hits[last_hit].from0InProt + (K - 1),
                               fi_count, current_fi, weighted))
            if self.debug:
                pw.write("after-call: ")
                self.display_hits(hits, pw)
            for i in range(last_hit + 1):
                if hits[i].fI == current_fi:
                    j = 0
                    while j < len(oi_counts) and oi_counts[j][1] != hits[i].oI:
                        j += 1
                    if j == len(oi_counts):
                        if len(oi_counts) == OI_BUFSZ:
                            j -= 1
                        else:
                            oi_counts.append([0, 0])
                        oi_counts[j][1] = hits[i].oI
                        oi_counts[j][0] = 1
                    else:
                        oi_counts[j][0] += 1
                    while j > 0 and oi_counts[j - 1][0] <= oi_counts[j][0]:
                        oi_counts[j - 1], oi_counts[j] = oi_counts[j], oi_counts[j - 1]
                        j -= 1
        n = len(hits)
        if n < 2:
            raise IndexError("hits.get(numHits-2)")   # Java would throw here
        if hits[n - 2].fI != current_fi and hits[n - 2].fI == hits[n - 1].fI:
            current_fi = hits[n - 1].fI
            hits[0] = hits[n - 2]
            hits[1] = hits[n - 1]
            del hits[2:]
        else:
            del hits[:]
        return current_fi

    # ---- KGJ:457-514
    def gather_hits(self, ln_dna, strand, frame, all_hits, function_array, oi_counts, pw):
        all_hits.sort(key=lambda h: h.from0InProt)       # stable, like Collections.sort
        hits = []
        current_fi = 0
        for ph in all_hits:
            avg_off_end = ph.avgOffFromEnd
            fi = ph.fI
            if self.debug:
                pw.write("HIT\t%d\t%d\t%d\t%d\t%s\t%d\n" % (
                    ph.from0InProt, 0, avg_off_end, fi, java_format_f(ph.functionWt, 3), ph.oI))
            if len(hits) > 0 and _i32(hits[-1].from0InProt + self.maxGap) < ph.from0InProt:
                if len(hits) >= self.minHits:
                    current_fi = self.process_set_of_hits(hits, function_array, current_fi, oi_counts, pw)
                else:
                    del hits[:]
            if not hits:
                current_fi = fi
            ok = (not self.orderConstraint) or len(hits) == 0
            if not ok:
                d = _i32(_i32(ph.from0InProt - hits[-1].from0InProt) - _i32(hits[-1].avgOffFromEnd - avg_off_end))
                ad = _i32(-d) if d < 0 else d            # Math.abs(int)
                ok = fi == hits[-1].fI and ad <= 20
            if ok:
                if len(hits) < MAX_HITS_PER_SEQ - 2:
                    hits.append(ph)
                    if self.debug:
                        pw.write("after-hit: ")
                        self.display_hits(hits, pw)
                if len(hits) > 1 and current_fi != fi and hits[-2].fI == hits[-1].fI:
                    current_fi = self.process_set_of_hits(hits, function_array, current_fi, oi_counts, pw)
        if len(hits) >= self.minHits:
            self.process_set_of_hits(hits, function_array, current_fi, oi_counts, pw)

    # ---- KGJ:516-524
    def tabulate_otu(self, current_id, length, oi_counts, pw):
        pw.write("OTU-COUNTS\t%s[%d]" % (current_id, length))
        for cnt, oi in oi_counts:
            pw.write("\t%d-%d" % (cnt, oi))
        pw.write("\n")
        self.otus.append([(c, o) for c, o in oi_counts])
        del oi_counts[:]

    # ---- KGJ:526-536
    def process_aa_seq(self, qid, prot_len, hit_cnts, function_array, pw):
        oi_counts = []
        pw.write("PROTEIN-ID\t%s\t%d\n" % (qid, prot_len))
        self._container = hit_cnts[(qid, "+", 0)]["id"]
        self.gather_hits(prot_len, "+", 0, hit_cnts[(qid, "+", 0)]["hits"], function_array, oi_counts, pw)
        self.tabulate_otu(qid, prot_len, oi_counts, pw)

    # ---- KGJ:538-558
    def process_seq(self, qid, contig_len, hit_cnts, function_array, pw):
        oi_counts = []
        pw.write("processing %s[%d]\n" % (qid, contig_len))
        for strand in "+-":
            for frame in range(3):
                pw.write("TRANSLATION\t%s\t%d\t%s\t%d\n" % (qid, contig_len, strand, frame))
                cnt = hit_cnts[(qid, strand, frame)]
                self._container = cnt["id"]
                self.gather_hits(contig_len, strand, frame, cnt["hits"], function_array, oi_counts, pw)
        self.tabulate_otu(qid, contig_len, oi_counts, pw)

    # ---- KGJ:900-922
    def add_kmers(self, qid, strand, frame, p_iseq, query_kmers, hit_cnts):
        cnt = {"key": (qid, strand, frame), "hits": [], "id": len(hit_cnts)}
        hit_cnts.append(cnt)
        for i in range(len(p_iseq) - K):
            value = encoded_kmer(p_iseq, i)
            if value < 0:
                continue
            query_kmers.append((value, cnt["id"], i))     # QueryKmer{value, hitCntId, protPos}

    # ---- KGJ:1051-1074
    def prepare_query(self, qid, sequence, query_kmers, hit_cnts):
        seq = sequence
        if self.aa:
            p_iseq = [to_amino_acid_off(c) for c in seq]
            self.add_kmers(qid, "+", 0, p_iseq, query_kmers, hit_cnts)
        else:
            ln = len(seq) // 3 + 1
            pseq = ["\0"] * ln
            p_iseq = [0] * ln
            for frame in range(3):
                translate(seq, frame, pseq, p_iseq)
                self.add_kmers(qid, "+", frame, p_iseq, query_kmers, hit_cnts)
            cs = rev_comp(seq)
            for frame in range(3):
                translate(cs, frame, pseq, p_iseq)
                self.add_kmers(qid, "-", frame, p_iseq, query_kmers, hit_cnts)

    # ---- KGJ:944-1034 (literal merge-join over the table stream)
    def lookup(self, stream: io.BytesIO, num_sigs, entry_size, query_kmers, hit_cnts):
        self.kmers_found = 0                                            # KGJ:956-957
        self.pos_count = 0
        self.processed = []                                             # (tenth, found-so-far) of every "Processed:" line
        fraction = 0                                                    # KGJ:963
        cur_hash = 0
        it = iter(query_kmers)
        cur = next(it, None)
        in_progress = {}
        while cur is not None or in_progress:
            needed = cur_hash
            if not in_progress:
                qk = cur
                needed = qk[0] % num_sigs
                in_progress[qk[0]] = [qk]
                cur = next(it, None)
            while cur is not None:
                qk = cur
                if qk[0] % num_sigs != needed:
                    break
                in_progress.setdefault(qk[0], []).append(qk)
                cur = next(it, None)
            if needed > cur_hash:
                to_skip = entry_size * (needed - cur_hash)                  # KGJ:991-994 skipBytesFully
                before = stream.tell()
                stream.seek(to_skip, io.SEEK_CUR)
                if stream.tell() > len(stream.getbuffer()) and self.gz:
                    # a GZIPInputStream skips by reading: it comes up short at the end of the data and skipBytesFully
                    # gives up (KGJ:1036-1049); a FileInputStream seeks past the end without complaint and the read
                    # below throws EOFException instead
                    raise SkipError("Error skipping %d bytes" % to_skip)
                assert stream.tell() == before + to_skip
                cur_hash = needed
            raw = stream.read(24)
            if len(raw) < 24:
                raise EOFError()
            which, otu_index, avg_from_end, function_index, function_wt = struct.unpack("<qiiif", raw)
            if which > MAX_ENCODED:
                in_progress.clear()
            elif which in in_progress:
                self.kmers_found += 1                                   # KGJ:1005
                for qk in in_progress.pop(which):
                    hit_cnts[qk[1]]["hits"].append(
                        Hit(otu_index, qk[2], avg_from_end, function_index, function_wt))
                    self.pos_count += 1                                 # KGJ:1014
            cur_hash += 1
            new_fraction = int(10.0 * (float(cur_hash) / float(num_sigs)))      # KGJ:1017-1024 (Java double arithmetic, (int) truncates)
            if new_fraction != fraction:
                fraction = new_fraction
                self.processed.append((fraction, self.kmers_found))
                self.info_lines.append("Processed: %d%%, time=0 ms., found-so-far=%d" % (fraction * 10, self.kmers_found))

    # ---- KGJ:742-820
    def run(self, table_bytes: bytes, function_array, fasta_text: str) -> str:
        pw = io.StringIO()
        num_sigs, entry_size, _version = struct.unpack("<qqq", table_bytes[:24])   # KGJ:933-935
        stream = io.BytesIO(table_bytes)
        stream.seek(24)
        hit_cnts = []
        query_id_to_len = {}
        query_kmers = []

        def cb(qid, seq, descr):
            self.prepare_query(qid, seq, query_kmers, hit_cnts)
            query_id_to_len[qid] = len(seq)

        read_fasta(fasta_text, cb)
        # KGJ:1076-1095
        def cmp(o1, o2):
            h1 = o1[0] % num_sigs; h2 = o2[0] % num_sigs
            if h1 != h2:
                return -1 if h1 < h2 else 1
            if o1[0] != o2[0]:
                return -1 if o1[0] < o2[0] else 1
            return 0
        query_kmers.sort(key=cmp_to_key(cmp))
        self.info_lines = ["Kmer-table info: numSigs=%d, entrySize=%d, version=%d" % (num_sigs, entry_size, _version)]
        try:
            self.lookup(stream, num_sigs, entry_size, query_kmers, hit_cnts)
            # KGJ:1031-1033 (debug only; not reached when the lookup ends in an exception)
            self.info_lines.append("Kmers found: %d (pos-count=%d)" % (self.kmers_found, self.pos_count))
        except EOFError:
            self.info_lines.append("Error: null")           # KGJ:799-802: EOFException() has no message; swallowed
        except SkipError as ex:
            self.info_lines.append("Error: " + str(ex))     # IllegalStateException("Error skipping N bytes"), KGJ:1045-1047
        by_key = {}
        for cnt in hit_cnts:                                # KGJ:805-809 (later container wins)
            by_key[cnt["key"]] = cnt
        for cnt in hit_cnts:
            for h in sorted(cnt["hits"], key=lambda h: h.from0InProt):
                self.hits.append((cnt["id"], h.from0InProt, h.oI, h.avgOffFromEnd, h.fI, h.functionWt))
        for qid, ln in query_id_to_len.items():
            if self.aa:
                self.process_aa_seq(qid, ln, by_key, function_array, pw)
            else:
                self.process_seq(qid, ln, by_key, function_array, pw)
        return pw.getvalue()


def _java_trim(s: str) -> str:
    """String.trim(): strips chars <= U+0020 at both ends."""
    a, b = 0, len(s)
    while a < b and s[a] <= " ":
        a += 1
    while b > a and s[b - 1] <= " ":
        b -= 1
    return s[a:b]


def _java_lines(text: str):
    """BufferedReader.readLine(): lines end at \\n, \\r or \\r\\n; no trailing empty line."""
    out = []
    i, n = 0, len(text)
    while i < n:
        j = i
        while j < n and text[j] not in "\r\n":
            j += 1
        out.append(text[i:j])
        if j < n and text[j] == "\r" and j + 1 < n and text[j + 1] == "\n":
            j += 1
        i = j + 1
    return out


def read_fasta(text: str, cb) -> None:
    """KGJ:1132-1192"""
    lines = _java_lines(text)
    pos = 0

    def read_line():
        nonlocal pos
        if pos >= len(lines):
            return None
        pos += 1
        return lines[pos - 1]

    str1 = None
    while True:
        prot_name = None
        prot_descr = None
        if str1 is None:
            str1 = read_line()
        while True:
            if str1 is None:
                break
            str2 = _java_trim(str1)
            if len(str2) > 1:
                if str2[0] == ">" and len(_java_trim(str2[1:])) > 0:
                    toks = [t for t in str2[1:].replace("\t", " ").split(" ") if t]   # StringTokenizer " \t"
                    prot_name = toks[0]
                    prot_descr = " ".join(toks[1:])
                    break
                raise ValueError("Wrong caption line: " + str2)
            str1 = read_line()
        if prot_name is None:
            break
        while True:
            str1 = read_line()
            if str1 is None or _java_trim(str1).startswith(">"):
                raise ValueError("No sequence for caption: " + prot_name)
            if len(_java_trim(str1)) > 0:
                break
        sb = []
        while True:
            sb.append(str1)
            str1 = read_line()
            if str1 is None or _java_trim(str1).startswith(">"):
                break
        prot_seq = "".join(sb)
        if len(prot_seq) == 0:
            raise ValueError("No sequence for caption: " + prot_name)
        cb(prot_name, prot_seq, prot_descr)


def load_indexed_array(text: str):
    """KGJ:345-369"""
    ret = []
    for line_pos, line in enumerate(_java_lines(text)):
        tab = line.index("\t") if "\t" in line else -1
        index = int(line[:tab]) if tab >= 0 else int(line[:-1])  # substring(0,-1) would throw in Java
        if line_pos != index:
            raise ValueError("Your index must be dense and in order (see line %d)" % line_pos)
        ret.append(line[tab + 1:])
    return ret
