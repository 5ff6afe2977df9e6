// kmer_guts_cli.cpp -- native host front end of the hot path: the reference's command line
// (KmerGutsJava.main, KGJ:560-654) and run() (KGJ:742-820) in C++ over the C ABI.
//
// "KGJ:n" = reference lib/src/kmergutsjava/KmerGutsJava.java line n.  The reference is compiled code (Java)
// and the build image has no JDK, so the host side above libkmerguts_hip.so is C++ here (and Python in
// kmer_guts_java.py; both produce the same bytes).  This file parses text and prints records; every number in
// the report comes from the GPU through kg_scan.  There is no CPU fallback.
//
//   kmer_guts -D DataDir [-q query.fasta[.gz]] [-o out.txt] [-a] [-d] [-m minHits] [-M minWeightedHits] [-O] [-g maxGap]
//
// Same flags, same quirks: -t / -l fall into "Unknown parameter" (KGJ:605-611); after a parse error the usage
// is printed and execution continues (KGJ:616-647); a missing -q is an error (new File(null), KGJ:647).
// Differences: no "Processed: NN%" lines (the table is not streamed); -d prints the info lines only.
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/kmerguts_hip.h"

namespace {

struct Fatal { std::string msg; };

[[noreturn]] void die(const std::string &m) { throw Fatal{m}; }

// ---- file input (plain or .gz, KGJ:347-352, 764-769) ----
std::vector<char> read_all(const std::string &path)
{
    std::vector<char> out;
    const bool gz = path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
    if (gz) {
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) die(path + " (No such file or directory)");
        gzbuffer(f, 1 << 20);
        std::vector<char> buf(8 << 20);
        int n;
        while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) out.insert(out.end(), buf.begin(), buf.begin() + n);
        gzclose(f);
        if (n < 0) die("error reading " + path);
    } else {
        FILE *f = path == "-" ? stdin : fopen(path.c_str(), "rb");
        if (!f) die(path + " (No such file or directory)");
        std::vector<char> buf(8 << 20);
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) out.insert(out.end(), buf.begin(), buf.begin() + n);
        if (f != stdin) fclose(f);
    }
    return out;
}

bool exists(const std::string &p)
{
    FILE *f = fopen(p.c_str(), "rb");
    if (f) fclose(f);
    return f != nullptr;
}

// ---- BufferedReader.readLine over a byte buffer: \n, \r or \r\n end a line ----
struct Lines {
    const char *p, *end;
    bool next(const char *&b, const char *&e)
    {
        if (p >= end) return false;
        b = p;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *stop = nl ? nl : end;
        const char *cr = (const char *)memchr(p, '\r', (size_t)(stop - p));      // a lone \r ends a line too
        if (cr) stop = cr;
        p = e = stop;
        if (p < end) {
            if (*p == '\r' && p + 1 < end && p[1] == '\n') p++;
            p++;
        }
        return true;
    }
};

// String.trim(): strips chars <= ' ' at both ends
void trim(const char *&b, const char *&e)
{
    while (b < e && (unsigned char)*b <= ' ') b++;
    while (e > b && (unsigned char)e[-1] <= ' ') e--;
}

// ---- loadIndexedArray (KGJ:345-369) ----
std::vector<std::string> load_indexed_array(const std::vector<char> &text)
{
    std::vector<std::string> out;
    Lines ln{text.data(), text.data() + text.size()};
    const char *b, *e;
    long pos = 0;
    while (ln.next(b, e)) {
        const char *tab = (const char *)memchr(b, '\t', (size_t)(e - b));
        if (!tab) die("String index out of range: -1");
        char *endp = nullptr;
        std::string num(b, tab);
        errno = 0;
        long idx = strtol(num.c_str(), &endp, 10);
        if (num.empty() || *endp || errno) die("For input string: \"" + num + "\"");
        if (idx != pos) die("Your index must be dense and in order (see line " + std::to_string(pos) + ")");
        out.emplace_back(tab + 1, e);
        pos++;
    }
    return out;
}

// ---- readFasta (KGJ:1132-1192): id = first token after '>', sequence lines concatenated untrimmed ----
struct Fasta {
    std::vector<std::string> ids;
    std::vector<uint8_t> seq;          // all sequences back to back
    std::vector<int64_t> off{0};
};

// the reference's reader over text[0, n); throws Fatal exactly where (and with the message) the reference throws
void read_fasta_range(const char *text, size_t n, Fasta &fa)
{
    Lines ln{text, text + n};
    const char *b = nullptr, *e = nullptr;
    bool have = false;                  // str1 carried over from the previous record
    fa.seq.reserve(n);
    for (;;) {
        std::string name;
        bool got_name = false;
        if (!have) have = ln.next(b, e);
        while (have) {
            const char *tb = b, *te = e;
            trim(tb, te);
            if (te - tb > 1) {
                const char *rb = tb + 1, *re = te;
                trim(rb, re);
                if (*tb == '>' && re > rb) {
                    // StringTokenizer(str2.substring(1), " \t").nextToken()
                    const char *q = tb + 1;
                    while (q < te && (*q == ' ' || *q == '\t')) q++;
                    const char *q2 = q;
                    while (q2 < te && *q2 != ' ' && *q2 != '\t') q2++;
                    name.assign(q, q2);
                    got_name = true;
                    break;
                }
                die("Wrong caption line: " + std::string(tb, te));
            }
            have = ln.next(b, e);
        }
        if (!got_name) return;
        for (;;) {
            have = ln.next(b, e);
            const char *tb = b, *te = e;
            if (have) trim(tb, te);
            if (!have || (te > tb && *tb == '>')) die("No sequence for caption: " + name);
            if (te > tb) break;
        }
        const size_t start = fa.seq.size();
        for (;;) {
            fa.seq.insert(fa.seq.end(), (const uint8_t *)b, (const uint8_t *)e);     // untrimmed (KGJ:1176)
            have = ln.next(b, e);
            if (!have) break;
            const char *tb = b, *te = e;
            trim(tb, te);
            if (te > tb && *tb == '>') break;
        }
        if (fa.seq.size() == start) die("No sequence for caption: " + name);
        fa.ids.push_back(std::move(name));
        fa.off.push_back((int64_t)fa.seq.size());
    }
}

// A line whose trimmed form starts with '>' ends the record before it whatever follows (KGJ:1163-1180), so the text
// can be cut in front of such lines and the pieces read independently: the records, their order and the first
// error in file order are those of one sequential pass.
void read_fasta(const char *text, size_t n, Fasta &fa)
{
    const unsigned hw = std::thread::hardware_concurrency();
    size_t n_thr = std::min<size_t>(16, hw ? hw : 4);
    const char *tv = getenv("KG_FASTA_THREADS");                 // tests force several pieces on small inputs
    if (tv && atoi(tv) > 0) n_thr = (size_t)atoi(tv);
    else if (n < (8u << 20)) n_thr = 1;
    std::vector<size_t> cut{0};
    for (size_t k = 1; k < n_thr; k++) {
        size_t p = n / n_thr * k;
        if (p <= cut.back()) continue;
        // the first line start at or after p whose first character > ' ' is '>'
        const char *q = (const char *)memchr(text + p - 1, '\n', n - p + 1);      // (a \r-only file is read by one thread)
        while (q) {
            const char *ls = q + 1, *c = ls;
            while (c < text + n && (unsigned char)*c <= ' ' && *c != '\n' && *c != '\r') c++;
            if (c < text + n && *c == '>') { cut.push_back((size_t)(ls - text)); break; }
            q = (const char *)memchr(ls, '\n', (size_t)(text + n - ls));
        }
        if (!q) break;
    }
    cut.push_back(n);
    const size_t parts = cut.size() - 1;
    if (parts == 1) { read_fasta_range(text, n, fa); return; }
    std::vector<Fasta> part(parts);
    std::vector<std::string> err(parts);
    std::vector<char> failed(parts, 0);
    {
        std::vector<std::thread> pool;
        for (size_t k = 0; k < parts; k++)
            pool.emplace_back([&, k]() {
                try { read_fasta_range(text + cut[k], cut[k + 1] - cut[k], part[k]); }
                catch (const Fatal &f) { failed[k] = 1; err[k] = f.msg; }
            });
        for (auto &t : pool) t.join();
    }
    // a piece that stops at an error still holds the records in front of it; the sequential reader would have thrown
    // there too, after the same records -- nothing is reported in that case, so only the message matters
    for (size_t k = 0; k < parts; k++)
        if (failed[k]) die(err[k]);
    size_t total = 0, n_ids = 0;
    std::vector<size_t> base(parts);
    for (size_t k = 0; k < parts; k++) { base[k] = total; total += part[k].seq.size(); n_ids += part[k].ids.size(); }
    fa.seq.resize(total);
    fa.ids.reserve(n_ids);
    fa.off.reserve(n_ids + 1);
    {
        std::vector<std::thread> pool;
        for (size_t k = 0; k < parts; k++)
            pool.emplace_back([&, k]() { if (!part[k].seq.empty()) memcpy(fa.seq.data() + base[k], part[k].seq.data(), part[k].seq.size()); });
        for (auto &t : pool) t.join();
    }
    for (size_t k = 0; k < parts; k++) {
        for (auto &id : part[k].ids) fa.ids.push_back(std::move(id));
        for (size_t r = 1; r < part[k].off.size(); r++) fa.off.push_back((int64_t)base[k] + part[k].off[r]);
    }
}

// the query text: a plain file is mapped, not copied; .gz and stdin are read into memory
struct Text {
    const char *p = nullptr;
    size_t n = 0;
    std::vector<char> owned;
    void *map = nullptr;
    ~Text() { if (map) munmap(map, n); }
};

void load_text(const std::string &path, Text &t)
{
    const bool gz = path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
    if (!gz && path != "-") {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) die(path + " (No such file or directory)");
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            close(fd);
            if (m != MAP_FAILED) { t.map = m; t.p = (const char *)m; t.n = (size_t)st.st_size; return; }
        } else close(fd);
    }
    t.owned = read_all(path);
    t.p = t.owned.data();
    t.n = t.owned.size();
}

// ---- String.format("%f") of a float: decimal digits of (double)v rounded HALF_UP (java.util.Formatter) ----
// For a float the only inputs on which HALF_UP and printf's round-half-even differ are exact ties,
// i.e. v * 2^(p+1) an odd integer.
int format_java_f(float v, int precision, char *buf, size_t bufsz)
{
    double d = (double)v;
    if (std::isnan(d)) return snprintf(buf, bufsz, "NaN");
    if (std::isinf(d)) return snprintf(buf, bufsz, d < 0 ? "-Infinity" : "Infinity");
    double ad = std::fabs(d);
    double scaled = std::ldexp(ad, precision + 1);
    if (precision >= 0 && precision <= 9 && scaled < 9.0e15 && scaled == std::floor(scaled) && std::fmod(scaled, 2.0) == 1.0) {
        uint64_t k = (uint64_t)scaled, p5 = 1, p10 = 1;
        for (int i = 0; i < precision; i++) { p5 *= 5; p10 *= 10; }
        if (k < UINT64_MAX / p5 - 1) {
            uint64_t units = (k * p5 + 1) / 2;
            return snprintf(buf, bufsz, "%s%llu.%0*llu", std::signbit(d) ? "-" : "", (unsigned long long)(units / p10), precision,
                            (unsigned long long)(units % p10));
        }
    }
    return snprintf(buf, bufsz, "%.*f", precision, d);
}

struct Options {
    bool aa = false, order_constraint = false, debug = false;
    int min_hits = 5, min_weighted_hits = 0, max_gap = 200;
    bool has_dir = false, has_query = false, has_out = false;
    std::string dir, query, out;
};

int parse_int(const char *s)
{
    if (!s) die("null");
    char *endp = nullptr;
    errno = 0;
    long v = strtol(s, &endp, 10);
    if (!*s || *endp || errno || v < INT32_MIN || v > INT32_MAX) die(std::string("For input string: \"") + s + "\"");
    return (int)v;
}

const char *kUsage[] = {
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
};

void parse_args(int argc, char **argv, Options &o)
{
    int i = 1;
    auto poll = [&]() -> const char * { return i < argc ? argv[i++] : nullptr; };
    try {
        while (i < argc) {
            std::string param = argv[i++];
            if (param.empty() || param[0] != '-') die("Parameter name should start from '-': " + param);
            param = param.substr(1);
            if (param.size() != 1) die("Unknown parameter: -" + param);
            switch (param[0]) {
            case 'a': o.aa = true; break;
            case 'd': o.debug = true; break;
            case 'm': o.min_hits = parse_int(poll()); break;
            case 'M': o.min_weighted_hits = parse_int(poll()); break;
            case 'O': o.order_constraint = true; break;
            case 'g': o.max_gap = parse_int(poll()); break;
            case 'D': { const char *v = poll(); o.has_dir = v != nullptr; if (v) o.dir = v; break; }
            case 'q': { const char *v = poll(); o.has_query = v != nullptr; if (v) o.query = v; break; }
            case 'o': { const char *v = poll(); o.has_out = v != nullptr; if (v) o.out = v; break; }
            case 't': (void)poll();  /* falls through, KGJ:605-607 */
            case 'l': { const char *v = poll(); if (!v) die("null"); (void)strtoll(v, nullptr, 10); }  /* falls through, KGJ:607-609 */
            default: die("Unknown parameter: -" + param);
            }
        }
        if (!o.has_dir) die("-D parameter is required");
    } catch (const Fatal &f) {
        printf("Error: %s\n", f.msg.c_str());
        for (const char *l : kUsage) printf("%s\n", l);
    }
}

struct Out {
    FILE *f;
    std::string buf;
    void flush() { if (!buf.empty()) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); } fflush(f); }
    void put(const std::string &s) { buf += s; if (buf.size() > (8u << 20)) flush(); }
};

void check(int rc)
{
    if (rc != KG_OK) die(std::string("libkmerguts_hip: ") + kg_last_error());
}

long long now_ms()
{
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace


#ifndef KG_CLI_NO_MAIN
int main(int argc, char **argv)
{
    Options o;
    parse_args(argc, argv, o);
    try {
        if (!o.has_dir) die("kmerTableDir is null");
        if (!o.has_query) die("queryFastaFile is null");          // KGJ:647: new File(null)
        const bool to_stdout = !o.has_out;
        Out out{to_stdout ? stdout : fopen(o.out.c_str(), "wb"), {}};
        if (!out.f) die(o.out + " (cannot open for writing)");
        auto info = [&](const std::string &m) {                    // printInfoLine, KGJ:891-898
            if (o.debug) out.put(m + "\n");
            if (!to_stdout) printf("%s\n", m.c_str());
        };
        {   // KGJ:108, 744-748: the canonical path of java.io.tmpdir, which is /tmp on Linux whatever $TMPDIR says
            char *canon = realpath("/tmp", nullptr);
            info(std::string("Temp. directory: ") + (canon ? canon : "/tmp"));
            free(canon);
        }

        std::string table = o.dir + "/kmer.table.mem_map", fidx = o.dir + "/function.index";
        if (exists(table + ".gz")) table += ".gz";                 // KGJ:750-753
        if (exists(fidx + ".gz")) fidx += ".gz";
        const std::vector<std::string> functions = load_indexed_array(read_all(fidx));

        kg_table *tab = nullptr;
        check(kg_table_open(table.c_str(), 0, &tab));             // plain or .gz: streamed to the device by the library

        long long t1 = now_ms();
        Fasta fa;
        {
            Text text;
            load_text(o.query, text);
            read_fasta(text.p, text.n, fa);
        }
        info("Preparation time: " + std::to_string(now_ms() - t1) + " ms.");

        // queryIdToLen / hitCnts are maps (KGJ:772, 805-809): a repeated id is reported once, at the place of its
        // first record, with the length and the hits of its last record
        const int64_t n = (int64_t)fa.ids.size();
        std::unordered_map<std::string, int64_t> last_of;
        std::vector<int64_t> first_order;
        for (int64_t k = 0; k < n; k++) {
            auto it = last_of.find(fa.ids[(size_t)k]);
            if (it == last_of.end()) { last_of.emplace(fa.ids[(size_t)k], k); first_order.push_back(k); }
            else it->second = k;
        }
        std::vector<int64_t> order;
        order.reserve(first_order.size());
        for (int64_t k : first_order) order.push_back(last_of[fa.ids[(size_t)k]]);

        long long t2 = now_ms();
        if (o.debug) {                                             // KGJ:951-954
            int64_t ns = 0, es = 0, ver = 0, occ = 0;
            check(kg_table_info(tab, &ns, &es, &ver, &occ));
            out.put("Kmer-table info: numSigs=" + std::to_string(ns) + ", entrySize=" + std::to_string(es) +
                    ", version=" + std::to_string(ver) + "\n");
        }
        kg_params p{};
        p.aa = o.aa; p.order_constraint = o.order_constraint; p.min_hits = o.min_hits;
        p.min_weighted_hits = o.min_weighted_hits; p.max_gap = o.max_gap; p.flags = 0;
        const int per = o.aa ? 1 : 6;
        // KG_CLI_BATCH_CHARS: test hook (smaller batches)
        const int64_t kMaxBatchChars = getenv("KG_CLI_BATCH_CHARS") ? std::max(1ll, atoll(getenv("KG_CLI_BATCH_CHARS"))) : 1500000000ll;   // below the ABI's 2^32-256 windows per call
        if (o.debug || !to_stdout) {
            // What lookup prints besides the records -- "Processed: NN%, time=..., found-so-far=K" once per tenth of the table
            // the merge-join visits (KGJ:1016-1025), with -d "Kmers found: N (pos-count=M)" (KGJ:1031-1033) -- and how run()
            // reports the stream's failure (KGJ:797-802).  All of it is about EVERY FASTA record (the reference's lookup fills the
            // containers of records that a later record of the same id shadows in the report too, KGJ:805-809) and precedes
            // the report, so the records are scanned once more here, hit records only, with KG_F_PROGRESS: the library notes
            // which table slots the reference's stream would have visited and counts the distinct k-mers found.
            kg_params pc = p;
            pc.flags = KG_F_SKIP_AGGREGATE | KG_F_PROGRESS;
            long long pos_count = 0;
            std::vector<kg_progress> prog;
            std::vector<uint32_t> slots;                           // (several batches only: a k-mer found in two counts once)
            int64_t a = 0;
            while (a < n) {
                int64_t b = a;
                while (b < n && (b == a || fa.off[(size_t)b + 1] - fa.off[(size_t)a] <= kMaxBatchChars)) b++;
                std::vector<int64_t> boff((size_t)(b - a) + 1);
                for (int64_t k = a; k <= b; k++) boff[(size_t)(k - a)] = fa.off[(size_t)k] - fa.off[(size_t)a];
                kg_result *res = nullptr;
                check(kg_scan(tab, &pc, fa.seq.data() + fa.off[(size_t)a], boff.data(), b - a, &res));
                kg_stats stc;
                check(kg_result_stats(res, &stc));
                kg_progress g;
                check(kg_result_progress(res, &g));
                prog.push_back(g);
                pos_count += stc.n_hits;
                if (!(a == 0 && b == n) && stc.n_hits) {
                    const uint32_t *hs = kg_result_hit_slots(res);
                    if (!hs) die(std::string("libkmerguts_hip: ") + kg_last_error());
                    slots.insert(slots.end(), hs, hs + stc.n_hits);
                }
                kg_result_free(res);
                a = b;
            }
            if (!prog.empty()) {
                int64_t first[11], found_upto[11], last = -1, beyond = -1, kmers_found = 0;
                bool walk_ran_off = false;
                for (int f = 0; f <= 10; f++) { first[f] = -1; found_upto[f] = 0; }
                for (const kg_progress &g : prog) {                // the join runs in slot order: minimum / maximum over the batches
                    for (int f = 0; f <= 10; f++)
                        if (g.first_visited[f] >= 0 && (first[f] < 0 || g.first_visited[f] < first[f])) first[f] = g.first_visited[f];
                    last = std::max(last, g.last_visited);
                    if (g.first_beyond >= 0 && (beyond < 0 || g.first_beyond < beyond)) beyond = g.first_beyond;
                    walk_ran_off = walk_ran_off || g.walk_ran_off;
                }
                if (prog.size() == 1) {
                    for (int f = 0; f <= 10; f++) found_upto[f] = prog[0].found_upto[f];
                    kmers_found = prog[0].kmers_found;
                } else {
                    std::sort(slots.begin(), slots.end());
                    slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
                    kmers_found = (int64_t)slots.size();
                    for (int f = 0; f <= 10; f++)
                        if (first[f] >= 0) found_upto[f] = std::upper_bound(slots.begin(), slots.end(), (uint32_t)first[f]) - slots.begin();
                }
                for (int f = 1; f <= 10; f++)                      // (tenth 0 is where the join starts: never a change)
                    if (first[f] >= 0)
                        info("Processed: " + std::to_string(f * 10) + "%, time=" + std::to_string(now_ms() - t2) + " ms., found-so-far=" +
                             std::to_string(found_upto[f]));
                const bool gz = table.size() > 3 && table.compare(table.size() - 3, 3, ".gz") == 0;
                if (walk_ran_off) {
                    // a query that walks off the end of the table makes the reference's stream throw EOFException: run() prints
                    // "Error: null" and carries on, "Kmers found" is not reached (KGJ:797-802)
                    info("Error: null");
                } else if (beyond >= 0) {
                    // a table stream shorter than numSigs records and a query whose home slot lies behind its end: the join skips to
                    // it -- a GZIPInputStream comes up short ("Error skipping N bytes", KGJ:1036-1049), a plain file seeks past its
                    // end and the read behind it throws EOFException
                    // (a home slot right AT the end of the stream is skipped to without trouble; the read there is what fails)
                    const long long skip = 24ll * (beyond - (last + 1));
                    info(gz && beyond > prog[0].stream_slots ? "Error: Error skipping " + std::to_string(skip) + " bytes" : std::string("Error: null"));
                } else if (o.debug) {
                    out.put("Kmers found: " + std::to_string(kmers_found) + " (pos-count=" + std::to_string(pos_count) + ")\n");
                }
            }
        }
        long long t_group = 0;
        size_t at = 0;
        char num[64];
        while (at < order.size()) {
            // one batch: the next records, in report order.  No repeated ids and everything in one batch (the
            // usual case): scan the parsed buffer in place; otherwise gather the batch's records into one buffer.
            std::vector<uint8_t> bseq;
            std::vector<int64_t> boff{0};
            size_t end = at;
            const bool in_place = at == 0 && order.size() == (size_t)n && fa.off.back() <= kMaxBatchChars;
            if (in_place) {
                end = order.size();
            } else {
                while (end < order.size()) {
                    const int64_t k = order[end];
                    const int64_t len = fa.off[(size_t)k + 1] - fa.off[(size_t)k];
                    if (end > at && boff.back() + len > kMaxBatchChars) break;
                    bseq.insert(bseq.end(), fa.seq.begin() + fa.off[(size_t)k], fa.seq.begin() + fa.off[(size_t)k + 1]);
                    boff.push_back(boff.back() + len);
                    end++;
                }
            }
            const uint8_t *sptr = in_place ? fa.seq.data() : bseq.data();
            const std::vector<int64_t> &boffr = in_place ? fa.off : boff;
            kg_result *res = nullptr;
            check(kg_scan(tab, &p, sptr, boffr.data(), (int64_t)(end - at), &res));
            const kg_call *calls = kg_result_calls(res);
            const int64_t *ccs = kg_result_container_call_start(res);
            const kg_otu *otu = kg_result_otu(res);
            if (!ccs || !otu) die(std::string("libkmerguts_hip: ") + kg_last_error());
            const kg_hit *hits = nullptr;                          // -d: hit records + what gatherHits did at each
            const int64_t *chs = nullptr;
            const uint8_t *ev = nullptr, *tail = nullptr;
            if (o.debug) {
                hits = kg_result_hits(res); chs = kg_result_container_hit_start(res);
                ev = kg_result_hit_events(res); tail = kg_result_container_tail_events(res);
                if (!hits || !chs || !ev || !tail) die(std::string("libkmerguts_hip: ") + kg_last_error());
            }
            if (at == 0) info("Lookup time: " + std::to_string(now_ms() - t2) + " ms.");
            long long t3 = now_ms();
            for (size_t j = 0; j < end - at; j++) {
                const std::string &id = fa.ids[(size_t)order[at + j]];
                const long long len = (long long)(boffr[j + 1] - boffr[j]);
                std::string r;
                if (o.aa) r += "PROTEIN-ID\t" + id + "\t" + std::to_string(len) + "\n";              // KGJ:529
                else r += "processing " + id + "[" + std::to_string(len) + "]\n";                    // KGJ:541
                for (int f = 0; f < per; f++) {
                    if (!o.aa)                                                                         // KGJ:545
                        r += "TRANSLATION\t" + id + "\t" + std::to_string(len) + "\t" + (f < 3 ? "+" : "-") + "\t" +
                             std::to_string(f % 3) + "\n";
                    const int64_t cont = (int64_t)j * per + f;
                    auto put_call = [&](int64_t c) {                                                  // KGJ:398-404
                        const kg_call &cl = calls[c];
                        if (cl.fI < 0 || (size_t)cl.fI >= functions.size())
                            die("Index: " + std::to_string(cl.fI) + ", Size: " + std::to_string(functions.size()));
                        format_java_f(cl.weightedHits, 6, num, sizeof num);
                        r += "CALL\t" + std::to_string(cl.start) + "\t" + std::to_string(cl.end) + "\t" + std::to_string(cl.count) +
                             "\t" + std::to_string(cl.fI) + "\t" + functions[(size_t)cl.fI] + "\t" + num + "\n";
                    };
                    if (!o.debug) {
                        for (int64_t c = ccs[cont]; c < ccs[cont + 1]; c++) put_call(c);
                        continue;
                    }
                    // -d stream (KGJ:470-473 HIT, 498-501 after-hit, 406-409 after-call, displayHits KGJ:376-383).
                    // Nothing is decided here: the event bytes say when the reference's hits list grew, was
                    // processed, kept its last two members or was emptied.
                    std::vector<int64_t> live;
                    int64_t nxt = ccs[cont];
                    auto show = [&](const char *tag) {
                        r += tag;
                        r += "hits: ";
                        for (int64_t i : live) {
                            format_java_f(hits[i].functionWt, 6, num, sizeof num);
                            r += std::to_string(hits[i].from0InProt) + "/" + num + "/" + std::to_string(hits[i].fI) + " ";
                        }
                        r += "\n";
                    };
                    auto reset = [&](bool called, bool keep2) {
                        if (called) { put_call(nxt++); show("after-call: "); }
                        if (keep2 && live.size() >= 2) live.erase(live.begin(), live.end() - 2);
                        else live.clear();
                    };
                    for (int64_t i = chs[cont]; i < chs[cont + 1]; i++) {
                        const kg_hit &h = hits[i];
                        const uint8_t e = ev[i];
                        format_java_f(h.functionWt, 3, num, sizeof num);
                        r += "HIT\t" + std::to_string(h.from0InProt) + "\t0\t" + std::to_string(h.avgOffFromEnd) + "\t" +
                             std::to_string(h.fI) + "\t" + num + "\t" + std::to_string(h.oI) + "\n";
                        if (e & KG_EV_RESET_BEFORE) reset(e & KG_EV_CALL_BEFORE, e & KG_EV_KEEP2_BEFORE);
                        if (e & KG_EV_ACCEPTED) { live.push_back(i); show("after-hit: "); }
                        if (e & KG_EV_RESET_AFTER) reset(e & KG_EV_CALL_AFTER, e & KG_EV_KEEP2_AFTER);
                        if (r.size() > (4u << 20)) { out.put(r); r.clear(); }
                    }
                    if (tail[cont] & KG_EV_TAIL_CALL) reset(true, false);
                    if (nxt != ccs[cont + 1]) die("event bytes and CALL records disagree");
                }
                r += "OTU-COUNTS\t" + id + "[" + std::to_string(len) + "]";                          // KGJ:518-522
                for (int k2 = 0; k2 < otu[j].n; k2++) r += "\t" + std::to_string(otu[j].count[k2]) + "-" + std::to_string(otu[j].oI[k2]);
                r += "\n";
                out.put(r);
            }
            t_group += now_ms() - t3;
            kg_result_free(res);
            at = end;
        }
        info("Grouping time: " + std::to_string(t_group) + " ms.");
        out.flush();
        if (!to_stdout) fclose(out.f);
        kg_table_close(tab);
    } catch (const Fatal &f) {
        fprintf(stderr, "Exception: %s\n", f.msg.c_str());
        return 1;
    }
    return 0;
}
#endif  // KG_CLI_NO_MAIN
