// Test harness (CPU only): the native front end's FASTA reader, cut into pieces and read by several threads, against
// the text the Python mirror's reader (a restatement of KGJ:1132-1192) produces.  Prints one line per record:
// "<id>\t<length>\t<fnv1a of the sequence bytes>", or "ERROR\t<message>".
#define KG_CLI_NO_MAIN
#include "../../kmergutsjava_amd/csrc/kmer_guts_cli.cpp"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    try {
        Fasta fa;
        Text text;
        load_text(argv[1], text);
        read_fasta(text.p, text.n, fa);
        for (size_t k = 0; k < fa.ids.size(); k++) {
            uint64_t h = 1469598103934665603ull;
            for (int64_t i = fa.off[k]; i < fa.off[k + 1]; i++) { h ^= fa.seq[(size_t)i]; h *= 1099511628211ull; }
            printf("%s\t%lld\t%016llx\n", fa.ids[k].c_str(), (long long)(fa.off[k + 1] - fa.off[k]), (unsigned long long)h);
        }
    } catch (const Fatal &f) {
        printf("ERROR\t%s\n", f.msg.c_str());
    }
    return 0;
}
