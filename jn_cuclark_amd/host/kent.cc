// kent -- the single-node front end, command-line compatible with the reference's
// app/kent.cpp:822-1050:
//   kent -c -O <reads> | -P <r1> <r2>  -R <result> [-b -k -t -n -d -g -s --tsk --extended --gzipped --verbose]
//   kent -a <database> <result.csv> [-o <file>]      abundance table (bin/getAbundance)
//   kent -m <f1> <f2> [...] [-o <file>]              merge abundance tables of split runs
//   kent -r [<abundance file>]                       plain-text report -> results/report.txt
//   kent -d <database>, kent -v                      database set-up / installation check
// What the reference does with `system()` and its scripts is kept where the scripts exist (an unmodified
// scripts/classify_metagenome.sh execs ../bin/cuCLARK-l, which is ours: INTEGRATION.md B); without the
// scripts, -c runs bin/cuCLARK-l directly with -T/-D taken from scripts/.settings or from the two options
// this build adds (-T <targets> -D <dbdir>), and -a runs bin/getAbundance directly
// (scripts/estimate_abundance.sh only forwards its arguments).  -m and -r are text processing with the
// reference's formats (app/kent.cpp:605-820); tests/golden/abundance/ holds what the reference's kent wrote.
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

bool exists_file(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
bool exists_dir(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }

// directory of this executable: bin/, next to scripts/ and to the other tools
std::string self_dir()
{
    char buf[4096];
    const ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
    if (n <= 0) return ".";
    buf[n] = 0;
    std::string p(buf);
    const size_t s = p.rfind('/');
    return s == std::string::npos ? "." : p.substr(0, s);
}

// as the reference: "~" and relative database paths are anchored at $HOME (app/kent.cpp:48-62)
std::string resolve_database_path(const std::string &path)
{
    if (path.empty()) return path;
    const char *home = getenv("HOME");
    if (path[0] == '~') return home ? std::string(home) + path.substr(1) : path;
    if (path[0] != '/' && home && !exists_dir(path)) return std::string(home) + "/" + path;
    return path;
}

std::string shell_quote(const std::string &v)
{
    std::string q = "'";
    for (char c : v) { if (c == '\'') q += "'\"'\"'"; else q += c; }
    return q + "'";
}

bool positive_int(const char *text, int &value)
{
    if (!text || !*text) return false;
    char *end = nullptr;
    const long v = strtol(text, &end, 10);
    if (*end != '\0' || v <= 0 || v > INT_MAX) return false;
    value = (int)v;
    return true;
}

bool to_double(const std::string &text, double &value)
{
    if (text.empty() || text == "-") return false;
    char *end = nullptr;
    value = strtod(text.c_str(), &end);
    return *end == '\0';
}

std::string percent(double v)
{
    std::ostringstream o;
    o << std::fixed << std::setprecision(2) << v;
    return o.str();
}

std::vector<std::string> cells(const std::string &line)       // plain split at commas, empty cells kept
{
    std::vector<std::string> out;
    std::string part;
    std::istringstream ss(line);
    while (std::getline(ss, part, ',')) out.push_back(part);
    return out;
}

// ---- -c ---------------------------------------------------------------------------------------------
struct Classify {
    std::string input, pair, result, sampling, targets, dbdir;
    bool paired = false, tsk = false, extended = false, gzipped = false, verbose = false, full = false;
    int batches = 32, k = -1, min_freq = -1, threads = -1, devices = -1, gap = -1;
};

int classify(const Classify &o)
{
    if (o.input.empty()) { std::cerr << "Input file not specified." << std::endl; return 1; }
    if (o.batches <= 0) { std::cerr << "Batch size must be a positive integer." << std::endl; return 1; }
    char cwdbuf[4096];
    if (!getcwd(cwdbuf, sizeof cwdbuf)) { std::cerr << "Failed to get current working directory." << std::endl; return 1; }
    const std::string cwd(cwdbuf);
    auto absolute = [&](const std::string &p) { return !p.empty() && p[0] != '/' ? cwd + "/" + p : p; };
    const std::string in1 = absolute(o.input);
    if (!exists_file(in1)) { std::cerr << "Input file not found: " << in1 << std::endl; return 1; }
    const std::string in2 = absolute(o.pair);
    if (o.paired && !exists_file(in2)) { std::cerr << "Paired input file not found: " << in2 << std::endl; return 1; }
    const std::string result = !o.result.empty() && o.result[0] == '/' ? o.result : cwd + "/results/" + o.result;

    std::string tail = o.paired ? " -P " + shell_quote(in1) + " " + shell_quote(in2) : " -O " + shell_quote(in1);
    tail += " -R " + shell_quote(result) + " -b " + std::to_string(o.batches);
    std::string opts;
    if (o.k > 0) opts += " -k " + std::to_string(o.k);
    if (o.min_freq >= 0) opts += " -t " + std::to_string(o.min_freq);
    if (o.threads > 0) opts += " -n " + std::to_string(o.threads);
    if (o.devices > 0) opts += " -d " + std::to_string(o.devices);
    if (o.gap > 0) opts += " -g " + std::to_string(o.gap);
    if (!o.sampling.empty()) opts += " -s " + shell_quote(o.sampling);
    if (o.tsk) opts += " --tsk";
    if (o.extended) opts += " --extended";

    std::string command;
    const bool direct = !o.targets.empty() || !exists_file("./scripts/classify_metagenome.sh");
    if (!direct) {
        // the reference's way: the wrapper script prepends -T/-D from scripts/.settings and execs ../bin/cuCLARK-l
        command = "cd scripts && ./classify_metagenome.sh" + tail + (o.full ? "" : " --light") + opts;
        if (o.gzipped) command += " --gzipped";
        if (o.verbose) command += " --verbose";
    } else {
        // no scripts around: the classifier itself (it reads gzip files directly, so --gzipped needs no copy)
        std::string targets = o.targets, dbdir = o.dbdir;
        if (targets.empty()) {
            std::ifstream st("scripts/.settings");
            std::string line;
            while (std::getline(st, line)) {
                if (line.compare(0, 3, "-T ") == 0) targets = line.substr(3);
                if (line.compare(0, 3, "-D ") == 0) dbdir = line.substr(3);
            }
        }
        if (targets.empty() || dbdir.empty()) {
            std::cerr << "Classification script not found: ./scripts/classify_metagenome.sh (and no -T <targets> -D <dbdir> given)" << std::endl;
            return 1;
        }
        const std::string exe = self_dir() + (o.full ? "/cuCLARK" : "/cuCLARK-l");
        if (!exists_file(exe)) { std::cerr << "Classifier not found: " << exe << std::endl; return 1; }
        mkdir((cwd + "/results").c_str(), 0755);
        command = shell_quote(exe) + " -T " + shell_quote(absolute(targets)) + " -D " + shell_quote(absolute(dbdir)) + tail + opts;
        if (o.verbose) command += " --verbose";
    }
    const int rc = system(command.c_str());
    if (rc != 0) { std::cerr << "Classification command failed with exit code " << rc << std::endl; return 1; }
    return 0;
}

// ---- -a ---------------------------------------------------------------------------------------------
int abundance(const std::string &db, const std::string &result, const std::string &output)
{
    if (db.empty()) { std::cerr << "Database path is empty." << std::endl; return 1; }
    if (result.empty()) { std::cerr << "Result file path is empty." << std::endl; return 1; }
    if (!exists_file(result)) {
        std::cerr << "Classification output not found: " << result << std::endl;
        std::cerr << "Make sure you provide the correct path to the .csv file produced by classification." << std::endl;
        return 1;
    }
    const std::string dir = resolve_database_path(db);
    if (!exists_dir(dir)) { std::cerr << "Database directory not found: " << dir << std::endl; return 1; }
    std::string tool = "./scripts/estimate_abundance.sh";
    if (!exists_file(tool)) tool = self_dir() + "/getAbundance";
    if (!exists_file(tool)) { std::cerr << "Abundance script not found: ./scripts/estimate_abundance.sh" << std::endl; return 1; }
    const std::string command = shell_quote(tool) + " -D " + shell_quote(dir) + " -F " + shell_quote(result) + " > " + shell_quote(output);
    const int rc = system(command.c_str());
    if (rc != 0) { std::cerr << "Abundance estimation failed with exit code " << rc << std::endl; return 1; }
    std::cout << "Abundance estimation completed successfully." << std::endl;
    return 0;
}

// ---- -m ---------------------------------------------------------------------------------------------
struct Entry {
    std::string name, taxid, lineage;
    long long count = 0;
};

bool parse_abundance(const std::string &path, std::vector<Entry> &entries, bool &has_lineage)
{
    std::ifstream in(path.c_str());
    if (!in) { std::cerr << "Failed to open abundance file: " << path << std::endl; return false; }
    std::string header;
    if (!std::getline(in, header)) { std::cerr << "Abundance file is empty: " << path << std::endl; return false; }
    has_lineage = header.find("Lineage") != std::string::npos;
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        const std::vector<std::string> p = cells(line);
        Entry e;
        if (has_lineage) {
            if (p.size() < 6) continue;
            e.name = p[0]; e.taxid = p[1]; e.lineage = p[2]; e.count = strtoll(p[3].c_str(), nullptr, 10);
        } else {
            if (p.size() < 5) continue;
            e.name = p[0]; e.taxid = p[1]; e.count = strtoll(p[2].c_str(), nullptr, 10);
        }
        entries.push_back(e);
    }
    return true;
}

int merge(const std::vector<std::string> &files, const std::string &output)
{
    for (const auto &f : files)
        if (!exists_file(f)) { std::cerr << "Abundance file not found: " << f << std::endl; return 1; }
    std::map<std::string, Entry> merged;              // by taxon id
    bool any_lineage = false;
    for (const auto &f : files) {
        std::vector<Entry> entries;
        bool lin = false;
        if (!parse_abundance(f, entries, lin)) return 1;
        any_lineage = any_lineage || lin;
        for (const Entry &e : entries) {
            auto it = merged.find(e.taxid);
            if (it == merged.end()) { merged[e.taxid] = e; continue; }
            it->second.count += e.count;
            if (it->second.name.empty() && !e.name.empty()) it->second.name = e.name;
            if (it->second.lineage.empty() && !e.lineage.empty()) it->second.lineage = e.lineage;
        }
    }
    if (merged.empty()) { std::cerr << "No entries found in any input file." << std::endl; return 1; }
    long long grand = 0, unknown = 0;
    bool has_unknown = false;
    Entry unk;
    std::vector<Entry> rows;
    for (const auto &kv : merged) {
        const Entry &e = kv.second;
        grand += e.count;
        if (e.taxid == "UNKNOWN" || e.name == "UNKNOWN") { unknown = e.count; unk = e; has_unknown = true; }
        else rows.push_back(e);
    }
    const long long classified = grand - unknown;
    std::sort(rows.begin(), rows.end(), [](const Entry &a, const Entry &b) { return a.name < b.name; });
    std::ofstream out(output.c_str());
    if (!out) { std::cerr << "Failed to open output file: " << output << std::endl; return 1; }
    out << (any_lineage ? "Name,TaxID,Lineage,Count,Proportion_All(%),Proportion_Classified(%)"
                        : "Name,TaxID,Count,Proportion_All(%),Proportion_Classified(%)") << std::endl;
    for (const Entry &e : rows) {
        out << e.name << "," << e.taxid;
        if (any_lineage) out << "," << e.lineage;
        out << "," << e.count << "," << percent(grand > 0 ? 100.0 * e.count / grand : 0.0) << ","
            << percent(classified > 0 ? 100.0 * e.count / classified : 0.0) << std::endl;
    }
    if (has_unknown) {
        out << unk.name << "," << unk.taxid;
        if (any_lineage) out << "," << unk.lineage;
        out << "," << unknown << "," << percent(grand > 0 ? 100.0 * unknown / grand : 0.0) << ",-" << std::endl;
    }
    std::cout << "Merged " << files.size() << " abundance files (" << grand << " total reads) -> " << output << std::endl;
    return 0;
}

// ---- -r ---------------------------------------------------------------------------------------------
int report(const std::string &file)
{
    if (!exists_file(file)) { std::cerr << "Abundance result file not found: " << file << std::endl; return 1; }
    std::ifstream in(file.c_str());
    if (!in) { std::cerr << "Failed to open " << file << std::endl; return 1; }
    std::string header;
    if (!std::getline(in, header)) { std::cerr << "Abundance result file is empty." << std::endl; return 1; }
    const std::string output = "results/report.txt";
    std::ofstream out(output.c_str());
    if (!out) { std::cerr << "Failed to open " << output << " for writing." << std::endl; return 1; }
    struct Line { std::string name; double all, classified; };
    std::vector<Line> rows;
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        const std::vector<std::string> p = cells(line);
        if (p.size() < 6 || p[0] == "UNKNOWN") continue;
        double a = 0, c = 0;
        if (!to_double(p[4], a) || !to_double(p[5], c)) continue;
        rows.push_back(Line{p[0], a, c});
    }
    if (rows.empty()) {
        out << "RESULT" << std::endl << "No classified pathogens found in " << file << "." << std::endl;
        std::cout << "Report written to " << output << std::endl;
        return 0;
    }
    std::sort(rows.begin(), rows.end(), [](const Line &a, const Line &b) {
        return a.classified == b.classified ? a.name < b.name : a.classified > b.classified;
    });
    out << "RESULT" << std::endl;
    out << "Your read contains these pathogens, the percentage of all input reads (including unclassified) "
           "that hit this taxon and the percentage among only the reads that got classified that hit this taxon." << std::endl;
    for (const Line &r : rows)
        out << "- " << r.name << ": " << percent(r.all) << "% among all, " << percent(r.classified) << "% among classified" << std::endl;
    std::cout << "Report written to " << output << std::endl;
    return 0;
}

// ---- -v, -d -----------------------------------------------------------------------------------------
int verify()
{
    std::cout << "========================================\n  CuCLARK Installation Verification\n========================================\n\n";
    const std::string bin = self_dir();
    bool ok = true;
    std::cout << "1. Checking binaries..." << std::endl;
    for (const char *b : {"cuCLARK", "cuCLARK-l", "getTargetsDef", "getAbundance", "kent"}) {
        const bool have = exists_file(bin + "/" + b);
        std::cout << "   " << (have ? "✓ " : "✗ ") << bin << "/" << b << (have ? "" : " (missing)") << std::endl;
        ok = ok && have;
    }
    const std::string lib = bin + "/../jn_cuclark_amd/libmcclark.so";
    const bool have_lib = exists_file(lib);
    std::cout << "   " << (have_lib ? "✓ " : "✗ ") << lib << (have_lib ? "" : " (missing: make -C jn_cuclark_amd/csrc)") << std::endl;
    ok = ok && have_lib;
    std::cout << "\n2. Checking database setup..." << std::endl;
    const bool settings = exists_file("scripts/.settings");
    std::cout << "   " << (settings ? "✓ Database configured (scripts/.settings exists)"
                                    : "⚠ Database not configured (run: kent -d <database_path>, or pass -T/-D to kent -c)") << std::endl;
    std::cout << "\n========================================\n";
    if (ok && settings) std::cout << "Status: READY ✓" << std::endl;
    else if (ok) std::cout << "Status: Installation complete, database not ready" << std::endl;
    else std::cout << "Status: INCOMPLETE" << std::endl;
    std::cout << "========================================" << std::endl;
    return ok ? 0 : 1;
}

int database(const std::string &path)
{
    if (path.empty()) { std::cerr << "Database path is empty." << std::endl; return 1; }
    if (exists_file("scripts/.settings")) {
        std::cerr << "Database is already configured (scripts/.settings exists)." << std::endl;
        std::cerr << "To reconfigure, you must first reset the database." << std::endl;
        return 1;
    }
    const std::string dir = resolve_database_path(path);
    if (!exists_dir(dir) || !exists_dir(dir + "/Custom")) {
        std::cerr << "Database directory (with a Custom/ folder of FASTA files) not found: " << dir << std::endl;
        std::cerr << "Database error, exiting the program." << std::endl;
        return 1;
    }
    // the taxonomy joins (accession -> taxid -> lineage) are the reference scripts' business; they need the NCBI dumps
    const std::string script = "./scripts/set_targets.sh";
    if (!exists_file(script)) { std::cerr << "Set targets script not found: " << script << std::endl; return 1; }
    const std::string command = "cd scripts && ./set_targets.sh " + shell_quote(dir) + " custom";
    const int rc = system(command.c_str());
    if (rc != 0) { std::cerr << "set_targets.sh failed with exit code " << rc << std::endl; return 1; }
    std::cout << "Database is ready." << std::endl;
    return 0;
}

void help(const char *argv0)
{
    std::cout << "Usage: " << argv0 << " [OPTIONS]\n\nOptions:\n"
              << "  -v, --verify              Verify installation status\n"
              << "  -d <database_path>        Setup database targets\n"
              << "  -c [OPTIONS]              Classify reads\n"
              << "     -O <file>              Single-end input reads (required unless -P)\n"
              << "     -P <file1> <file2>     Paired-end input reads\n"
              << "     -R <file>              Results output file (required)\n"
              << "     -b <int>               Number of batches (default: 32)\n"
              << "     -k <int>               K-mer length, 2-32 (default: 31)\n"
              << "     -t <int>               Min k-mer frequency in targets (default: 0)\n"
              << "     -n <int>               Number of threads\n"
              << "     -d <int>               Number of devices (GPUs)\n"
              << "     -g <int>               Gap/non-overlapping k-mers for cuCLARK-l (default: 4)\n"
              << "     -s <factor>            Sampling factor (cuCLARK only)\n"
              << "     --tsk                  Target-specific k-mer files (detailed DB creation)\n"
              << "     --extended             Extended results output\n"
              << "     --gzipped              Input files are gzipped\n"
              << "     --verbose              Verbose diagnostic output\n"
              << "     -T <targets> -D <dir>  (this build) run bin/cuCLARK-l directly, without scripts/.settings\n"
              << "     --full                 (this build) the full-size table (bin/cuCLARK) instead of cuCLARK-l\n"
              << "  -a <database> <result> [-o <output>]\n"
              << "                            Estimate abundance (default output: results/abundance_result.csv)\n"
              << "  -m <f1> <f2> [f3...]      Merge abundance files from split runs\n"
              << "     -o <file>              Output file (default: results/abundance_merged.csv)\n"
              << "  -r [<abundance_file>]      Generate report (default: results/abundance_result.csv)\n"
              << "  -h, --help                Show this help" << std::endl;
}

std::string in_results(const std::string &f) { return f.find('/') == std::string::npos ? "results/" + f : f; }

} // namespace

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::cerr << "Usage: " << argv[0] << " [OPTIONS]" << std::endl;
        std::cerr << "Options: -h, --help, -v/--verify, -d <database_path>, -c -O <fastq> -R <result> [options], -a <database> <result> [-o <output>], -m <f1> <f2> [...], -r [<abundance_file>]" << std::endl;
        return 1;
    }
    const std::string arg = argv[1];
    if (arg == "-h" || arg == "--help") { help(argv[0]); return 0; }
    if (arg == "-v" || arg == "--verify") return verify();
    if (arg == "-d") {
        if (argc < 3) { std::cerr << "Missing database path for -d option." << std::endl; return 1; }
        return database(argv[2]);
    }
    if (arg == "-c") {
        Classify o;
        bool seen_in = false, seen_res = false;
        for (int i = 2; i < argc; ++i) {
            const std::string a(argv[i]);
            auto need_int = [&](int &v, const char *name) {
                if (i + 1 >= argc || !positive_int(argv[i + 1], v)) { std::cerr << "Missing or invalid argument for " << name << std::endl; std::exit(1); }
                ++i;
            };
            if (a == "-O") { if (i + 1 >= argc) { std::cerr << "Missing argument for -O" << std::endl; return 1; } o.input = argv[++i]; o.paired = false; seen_in = true; }
            else if (a == "-P") { if (i + 2 >= argc) { std::cerr << "-P requires two filenames" << std::endl; return 1; } o.input = argv[++i]; o.pair = argv[++i]; o.paired = true; seen_in = true; }
            else if (a == "-R") { if (i + 1 >= argc) { std::cerr << "Missing argument for -R" << std::endl; return 1; } o.result = argv[++i]; seen_res = true; }
            else if (a == "-b") need_int(o.batches, "-b");
            else if (a == "-k") need_int(o.k, "-k");
            else if (a == "-t") {
                if (i + 1 >= argc) { std::cerr << "Missing argument for -t" << std::endl; return 1; }
                char *end = nullptr;
                const long v = strtol(argv[++i], &end, 10);
                if (*end != '\0' || v < 0 || v > INT_MAX) { std::cerr << "Invalid argument for -t (must be a non-negative integer)" << std::endl; return 1; }
                o.min_freq = (int)v;
            }
            else if (a == "-n") need_int(o.threads, "-n");
            else if (a == "-d") need_int(o.devices, "-d");
            else if (a == "-g") need_int(o.gap, "-g");
            else if (a == "-s") { if (i + 1 >= argc) { std::cerr << "Missing argument for -s" << std::endl; return 1; } o.sampling = argv[++i]; }
            else if (a == "-T") { if (i + 1 >= argc) { std::cerr << "Missing argument for -T" << std::endl; return 1; } o.targets = argv[++i]; }
            else if (a == "-D") { if (i + 1 >= argc) { std::cerr << "Missing argument for -D" << std::endl; return 1; } o.dbdir = argv[++i]; }
            else if (a == "--tsk") o.tsk = true;
            else if (a == "--extended") o.extended = true;
            else if (a == "--gzipped") o.gzipped = true;
            else if (a == "--verbose") o.verbose = true;
            else if (a == "--full") o.full = true;
            else {
                std::cerr << "Unknown classify option: " << a << std::endl;
                std::cerr << "Usage: " << argv[0] << " -c -O <fastq> -R <result> [options]" << std::endl;
                return 1;
            }
        }
        if (!seen_in) { std::cerr << "Classification requires -O <fastq> or -P <file1> <file2>" << std::endl; return 1; }
        if (!seen_res) { std::cerr << "Classification requires -R <resultFile>" << std::endl; return 1; }
        return classify(o);
    }
    if (arg == "-a") {
        if (argc < 4) {
            std::cerr << "Usage: " << argv[0] << " -a <database_path> <result_file> [-o <output_file>]" << std::endl;
            std::cerr << "  <result_file> is the .csv file produced by classification (e.g. results/result.csv)" << std::endl;
            return 1;
        }
        std::string out = "results/abundance_result.csv";
        for (int i = 4; i < argc; ++i)
            if (std::string(argv[i]) == "-o" && i + 1 < argc) out = in_results(argv[++i]);
        return abundance(argv[2], argv[3], out);
    }
    if (arg == "-m") {
        std::vector<std::string> files;
        std::string out = "results/abundance_merged.csv";
        for (int i = 2; i < argc; ++i) {
            const std::string a(argv[i]);
            if (a == "-o") { if (i + 1 >= argc) { std::cerr << "Missing argument for -o" << std::endl; return 1; } out = in_results(argv[++i]); }
            else files.push_back(a);
        }
        if (files.size() < 2) {
            std::cerr << "Usage: " << argv[0] << " -m <file1> <file2> [file3 ...] [-o <output>]" << std::endl;
            std::cerr << "At least 2 abundance files are required." << std::endl;
            return 1;
        }
        return merge(files, out);
    }
    if (arg == "-r") return report(argc > 2 ? in_results(argv[2]) : "results/abundance_result.csv");
    std::cerr << "Unknown argument: " << arg << std::endl;
    std::cerr << "Usage: " << argv[0] << " -v | -d <database_path> | -c -O <fastq> -R <result> [options] | -a <database_path> <result_file> [-o <output>] | -m <f1> <f2> [...] | -r [<abundance_file>]" << std::endl;
    return 1;
}
