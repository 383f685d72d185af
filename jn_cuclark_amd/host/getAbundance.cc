// getAbundance -- per-taxon counts from the per-read CSV of the classifier (the consumer of
// the CSV; no GPU work).
//
// Same command line, stdout table and side files as the reference tool behind scripts/estimate_abundance.sh
// (src/getAbundance.cc:151-579):
//     getAbundance [-c <minConfidence>] [-g <minGamma>] [-a <minAbundance%>] [--highconfidence] [--krona] [--mpa]
//                  [-D <database dir>] -F <result1.csv> [<result2.csv> ...]
// A read counts for its Assignment column if Gamma >= minGamma and Confidence >= minConfidence (:339-342),
// otherwise for UNKNOWN.  With -D the taxonomy under <dir>/taxonomy/{nodes,names}.dmp gives every taxon its
// scientific name and its lineage (superkingdom;phylum;class;order;family;genus).  Output rows are sorted by
// name; proportions print with the stream's default precision (6 significant digits), as the reference's.
//
// Written against the reference's OUTPUT, not its code: tests/golden/abundance/ holds tables the reference
// tool printed for a toy taxonomy, and tests/test_abundance.py compares byte for byte.  Where the reference's
// behaviour is an accident of its implementation it is kept and named below (names.dmp is read only until
// every label has a name; the lineage-name lookup of a row that named a label uses that NAME as the id).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

// fields of a line; runs of separators count as one (empty fields vanish), as the reference's tokenizer
std::vector<std::string> split(const std::string &line, const char *seps)
{
    std::vector<std::string> out;
    std::string cur;
    for (char ch : line) {
        if (std::strchr(seps, ch) && ch != '\0') {
            if (!cur.empty()) { out.push_back(cur); cur.clear(); }
        } else {
            cur.push_back(ch);
        }
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

bool read_line(std::ifstream &f, std::string &line)
{
    if (!std::getline(f, line)) { line.clear(); return false; }
    return true;
}

struct Node {
    uint32_t parent = 0;
    uint8_t rank = 255;       // 0 species .. 6 superkingdom, 7 root
};

constexpr int NB = 8;

struct Step {                 // one level of a lineage: the taxon at that rank (0 = none)
    uint32_t id = 0;
    bool set = false;
};

typedef std::unordered_map<uint32_t, Node> Tree;

const Node &node_of(const Tree &t, uint32_t id)
{
    static const Node none;
    auto it = t.find(id);
    return it == t.end() ? none : it->second;
}

void load_nodes(const std::string &path, Tree &tree)
{
    std::ifstream f(path.c_str());
    if (!f) { std::cerr << "Failed to open " << path << std::endl; std::exit(-1); }
    static const std::map<std::string, uint8_t> ranks = {
        {"species", 0}, {"genus", 1}, {"family", 2}, {"order", 3}, {"class", 4}, {"phylum", 5}, {"superkingdom", 6}, {"root", 7}};
    std::cerr << "Loading nodes of taxonomy tree... ";
    std::string line;
    while (read_line(f, line)) {
        const std::vector<std::string> e = split(line, " |\t");
        if (e.size() < 3) continue;
        Node &n = tree[(uint32_t)std::atoi(e[0].c_str())];
        n.parent = (uint32_t)std::atoi(e[1].c_str());
        auto it = ranks.find(e[2]);
        // "species group" / "species subgroup" tokenise as species + group...: not a rank of the lineage
        if (it != ranks.end() && (e.size() == 3 || e[3].find("group") == std::string::npos)) n.rank = it->second;
    }
    std::cerr << "done\n";
}

// the first taxon met at each rank on the way to the root; false for an id the tree does not know
bool lineage_of(const Tree &tree, uint32_t taxid, Step (&line)[NB])
{
    for (auto &s : line) s = Step();
    uint32_t it = taxid;
    if (node_of(tree, it).parent == 0) return false;
    for (size_t guard = 0; guard < 1000; guard++) {
        const Node &n = node_of(tree, it);
        if (n.parent == 1) {
            line[NB - 1].id = 1;
            if (line[NB - 2].id == 0) { line[NB - 2].id = it; line[NB - 2].set = true; }     // no superkingdom on the way: the top-most taxon stands in
            break;
        }
        if (n.rank < NB && !line[n.rank].set) { line[n.rank].set = true; line[n.rank].id = it; }
        it = n.parent;
    }
    return true;
}

std::string mpa_name(const std::string &n)
{
    std::string r = n;
    for (char &c : r) if (c == ' ') c = '_';
    return r;
}

struct Row {
    std::string name, taxid;
    size_t count = 0;
    std::vector<Step> lineage;          // ranks 0..6
    bool operator<(const Row &o) const { return name < o.name; }
};

[[noreturn]] void usage()
{
    std::cerr << " -c <minConfidenceScore> -g <minGamma> -D <Directory_Path> -F <result1>.csv <result2>.csv ... <result_n>.csv -a <minAbundance> ... \n"
              << "\nDefinition of parameters: \n\n"
              << "-c <minConfidenceScore>   count only assignments with a confidence score of at least this value (0.5 .. 1.0, default 0.5);\n"
              << "                          the others are counted as UNKNOWN.\n"
              << "-g <minGamma>             the same for the gamma score (0 .. 1.0, default 0).\n"
              << "-D <Directory_Path>       the database directory given to set_targets.sh: scientific names and lineages are\n"
              << "                          loaded from <Directory_Path>/taxonomy/{nodes,names}.dmp.\n"
              << "-F <result1>.csv ...      result file(s) of the classifier, all produced in the same mode.\n"
              << "-a <minAbundance(%)>      print only estimations of at least this percentage (0 .. 100).\n"
              << "--highconfidence          the same as '-c 0.75 -g 0.03'.\n"
              << "--krona                   also write results.krn (taxon id, taxon id, count) for Krona's ktImportTaxonomy -m 3.\n"
              << "--mpa                     also write results.mpa (MetaPhlAn's two-column format).\n"
              << std::endl;
    std::exit(1);
}

} // namespace

int main(int argc, char **argv)
{
    if (argc < 3) usage();
    int f_begin = -1, f_end = -1, i_dir = -1;
    double min_conf = 0.5, min_gamma = 0, min_ab = 0;
    bool krona = false, mpa = false;
    for (int t = 1; t < argc; t++) {
        const std::string p(argv[t]);
        if (p == "--highconfidence" || p == "--hc") { min_conf = 0.75; min_gamma = 0.03; continue; }
        if (p == "--krona" || p == "--Krona" || p == "--KRONA") { krona = true; continue; }
        if (p == "--mpa" || p == "--MPA" || p == "--Mpa") { mpa = true; continue; }
        if (p == "-c") {
            if (++t >= argc) { std::cerr << "Please provide a minimum for the confidence score." << std::endl; return 1; }
            min_conf = std::atof(argv[t]);
            if (min_conf < 0.5 || min_conf > 1) { std::cerr << "Please provide a minimum confidence score between 0.5 and 1." << std::endl; return 1; }
            continue;
        }
        if (p == "-g") {
            if (++t >= argc) { std::cerr << "Please provide a minimum value for the gamma." << std::endl; return 1; }
            min_gamma = std::atof(argv[t]);
            if (min_gamma < 0 || min_gamma > 1) { std::cerr << "Please provide a minimum Gamma score between 0 and 1." << std::endl; return 1; }
            continue;
        }
        if (p == "-a") {
            if (++t >= argc) { std::cerr << "Please provide a minimum value for the abundance." << std::endl; return 1; }
            min_ab = std::atof(argv[t]);
            if (min_ab < 0 || min_ab > 100) { std::cerr << "Please provide a minimum abundance between 0 and 100." << std::endl; return 1; }
            continue;
        }
        if (p == "-D") {
            if (++t >= argc) { std::cerr << "Please provide the directory path containing the database." << std::endl; return 1; }
            i_dir = t;
            continue;
        }
        if (p == "-F") {
            if (++t >= argc) { std::cerr << "Please provide one (or several) CLARK output file(s) (CSV format)." << std::endl; return 1; }
            f_begin = t;
            while (++t != argc && argv[t][0] != '-') {}
            f_end = t;
            t--;
            continue;
        }
        std::cerr << "Failed to recognize option: " << argv[t] << std::endl;
        return 1;
    }
    if (f_begin < 0) { std::cerr << "Please provide one (or several) CLARK output file(s) (CSV format)." << std::endl; return 1; }

    // ---- count the assignments ----------------------------------------------------------------------
    std::map<std::string, uint32_t> index_of;                 // label -> row, in order of first appearance
    std::vector<size_t> counts;
    std::vector<std::string> labels, names;
    size_t total = 0, col = 0;
    for (int fi = f_begin; fi < f_end; fi++) {
        std::ifstream f(argv[fi]);
        if (!f) { std::cerr << "Failed to open " << argv[fi] << std::endl; return 1; }
        std::string line;
        read_line(f, line);                                     // header
        if (fi == f_begin) {
            const std::vector<std::string> h = split(line, ",\t\r");
            if (h.size() < 3) {
                std::cerr << "Failed to extract all data from the file: " << argv[fi] << ". The file does not seem to be a CLARK results file." << std::endl;
                return 1;
            }
            col = h.size() == 3 ? 2 : h.size() - 3;             // Assignment: third from the end (Score and Confidence follow)
        }
        std::cerr << "\rFile: " << argv[fi] << "    ";
        while (read_line(f, line)) {
            std::vector<std::string> e = split(line, ",\t\r");
            total++;
            if (e.size() <= col) continue;                       // (the reference indexes blindly)
            bool ok = true;
            if (e.size() > 3 && col >= 1 && col + 2 < e.size())
                ok = std::atof(e[col - 1].c_str()) >= min_gamma && std::atof(e[col + 2].c_str()) >= min_conf;
            const std::string label = ok ? e[col] : "NA";
            auto it = index_of.find(label);
            if (it == index_of.end()) {
                index_of[label] = (uint32_t)labels.size();
                counts.push_back(1);
                labels.push_back(label);
                names.push_back(label);
            } else {
                counts[it->second]++;
            }
        }
    }
    std::cerr << "\n";

    // ---- names and lineages ---------------------------------------------------------------------------
    std::vector<std::vector<Step>> lineages;
    std::map<uint32_t, std::string> taxon_name;               // every taxon that appears in a lineage
    if (i_dir > 0) {
        Tree tree;
        load_nodes(std::string(argv[i_dir]) + "/taxonomy/nodes.dmp", tree);
        lineages.resize(labels.size());
        std::cerr << "Start retrieving lineage for each target identified (" << labels.size() << ")... ";
        for (size_t i = 0; i < labels.size(); i++) {
            if (labels[i] == "NA") continue;
            Step line[NB];
            if (!lineage_of(tree, (uint32_t)std::atoi(labels[i].c_str()), line)) {
                std::cerr << "\nFailed to identify " << labels[i] << ": Unknown taxonomy id given the provided taxonomy database." << std::endl;
                labels[i] = "NA";
                names[i] = "NA";
                continue;
            }
            for (int t = 0; t < NB - 1; t++) {
                if (line[t].set && !taxon_name.count(line[t].id)) taxon_name[line[t].id] = "";
                lineages[i].push_back(line[t]);
            }
        }
        std::cerr << "done." << std::endl;

        const std::string npath = std::string(argv[i_dir]) + "/taxonomy/names.dmp";
        std::ifstream f(npath.c_str());
        if (!f) {
            std::cerr << "Failed to open " << npath << std::endl;
            std::cerr << "The program will estimates abundance per taxonomy id." << std::endl;
        } else {
            std::cerr << "Retrieving scientific names from taxonomy tree... ";
            size_t named = 0;
            std::string line;
            // kept from the reference: reading stops once every label has been named, whatever lineage names
            // are still missing by then
            while (named < labels.size() && read_line(f, line)) {
                const std::vector<std::string> e = split(line, "|");
                if (e.empty()) continue;
                std::vector<std::string> id = split(e[0], "\t");
                if (id.empty()) continue;
                const bool scientific = e.size() > 3 && e[3].find("scientific name") != std::string::npos;
                std::string key = id[0];
                auto it = index_of.find(key);
                if (it != index_of.end() && scientific) {
                    named++;
                    const std::vector<std::string> nm = split(e[1], "\t");
                    names[it->second] = nm.empty() ? "" : nm[0];
                    key = nm.empty() ? "" : nm[0];             // kept from the reference: the lineage lookup below then uses the NAME
                }
                auto il = taxon_name.find((uint32_t)std::atoi(key.c_str()));
                if (il != taxon_name.end() && scientific) {
                    const std::vector<std::string> nm = split(e[1], "\t");
                    il->second = nm.empty() ? "" : nm[0];
                }
            }
            std::cerr << "done." << std::endl;
        }
    }

    std::vector<Row> rows(labels.size());
    for (size_t t = 0; t < labels.size(); t++) {
        rows[t].taxid = labels[t];
        rows[t].name = names[t];
        rows[t].count = counts[t];
        if (rows[t].name == "NA") continue;
        if (!lineages.empty()) rows[t].lineage = lineages[t];
    }
    std::sort(rows.begin(), rows.end());

    // ---- the table -------------------------------------------------------------------------------------
    if (i_dir < 0) std::cout << "Name,TargetID,";
    else std::cout << "Name,TaxID,Lineage";
    std::cout << ",Count,Proportion_All(%),Proportion_Classified(%)" << std::endl;
    size_t unknown = 0;
    for (const Row &r : rows) if (r.name == "NA") unknown += r.count;
    for (const Row &r : rows) {
        if (r.name == "NA") continue;
        const double a = 100 * ((double)r.count) / ((double)total);
        const double a2 = 100 * ((double)r.count) / ((double)(total - unknown));
        if (a < min_ab) continue;
        std::cout << r.name << "," << r.taxid << ",";
        if (!r.lineage.empty()) {
            const size_t len = r.lineage.size();
            std::cout << taxon_name[r.lineage[len - 1].id];
            for (size_t u = len - 2; u > 0; u--) std::cout << ";" << taxon_name[r.lineage[u].id];
            std::cout << ",";
        }
        std::cout << r.count << "," << a << "," << a2 << std::endl;
    }
    {
        const double a = 100 * ((double)unknown) / ((double)total);
        if (a >= min_ab) {
            if (i_dir > 0) std::cout << "UNKNOWN,UNKNOWN,UNKNOWN," << unknown << "," << a << ",-" << std::endl;
            else std::cout << "UNKNOWN,UNKNOWN," << unknown << "," << a << ",-" << std::endl;
        }
    }

    if (krona) {
        std::ofstream out("results.krn", std::ios::binary);
        for (const Row &r : rows)
            if (r.name != "NA") out << r.taxid << " \t " << r.taxid << " \t " << r.count << std::endl;
    }
    if (mpa) {
        static const char *prefix[] = {"s__", "g__", "f__", "o__", "c__", "p__", "d__"};
        std::ofstream out("results.mpa", std::ios::binary);
        std::map<uint32_t, int> seen;
        // every taxon above species that some row's lineage holds, top rank first, with the reads at or below it
        for (size_t t = NB - 1; t > 0; t--) {
            for (size_t r = 0; r < rows.size(); r++) {
                if (rows[r].lineage.size() <= t || !rows[r].lineage[t].set) continue;
                const uint32_t taxon = rows[r].lineage[t].id;
                if (seen.count(taxon)) continue;
                seen[taxon] = 1;
                long reads = (long)rows[r].count;
                const size_t len = rows[r].lineage.size();
                out << prefix[len - 1] << mpa_name(taxon_name[rows[r].lineage[len - 1].id]);
                for (size_t v = len - 2; v >= t; v--) {
                    if (taxon_name[rows[r].lineage[v].id] != "") out << "|" << prefix[v] << mpa_name(taxon_name[rows[r].lineage[v].id]);
                    if (v == 0) break;
                }
                for (size_t s = 0; s < rows.size(); s++) {
                    if (r == s || rows[s].lineage.size() <= t) continue;
                    if (rows[s].lineage[t].id == taxon) reads += (long)rows[s].count;
                }
                out << "\t" << reads << std::endl;
            }
        }
        for (const Row &r : rows) {
            if (r.name == "NA") continue;
            const size_t len = r.lineage.size();
            if (len < 2) continue;                                // (no taxonomy loaded: nothing to print above the species)
            out << prefix[len - 1] << mpa_name(taxon_name[r.lineage[len - 1].id]);
            for (size_t v = len - 2; v > 0; v--)
                if (taxon_name[r.lineage[v].id] != "") out << "|" << prefix[v] << mpa_name(taxon_name[r.lineage[v].id]);
            out << "|" << prefix[0] << mpa_name(r.name);
            out << "\t" << r.count << std::endl;
        }
    }
    return 0;
}
