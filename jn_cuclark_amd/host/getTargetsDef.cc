// getTargetsDef -- files-to-taxonomy table -> targets definition (BASELINE config 1 plumbing).
//
// Same command line, output and side file as the reference tool that scripts/set_targets.sh:117
// runs (src/getTargetsDef.cc:38-96):
//     getTargetsDef <FilesToTaxIDs> [rank 0..5]  > targets.txt
// Input rows: <file> <taxid> <species> <genus> <family> <order> <class> <phylum>, fields separated by
// tabs, commas or blanks.  A row whose taxid is -1 is reported in files_excluded.txt (created even
// when empty, header line before the first entry); a row whose id at the wanted rank is UNKNOWN
// is dropped silently; every other row becomes "<file>\t<id at rank>".  With no rank argument the
// reference reads column 2+1 (genus) although its usage text promises species: kept, callers
// always pass the rank (set_targets.sh:117).
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

static std::vector<std::string> fields(const std::string &line)
{
    std::vector<std::string> out;
    std::string cur;
    for (char ch : line) {
        if (ch == '\t' || ch == ',' || ch == ' ') {
            if (!cur.empty()) { out.push_back(cur); cur.clear(); }
        } else {
            cur.push_back(ch);
        }
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::cerr << "Usage: " << argv[0]
                  << " <FilestoTaxIDs>, option: <Rank: 0,1,2,3,4,5>, 0 for species, 1 for genus, ..., 5 for phylum. Default is species."
                  << std::endl;
        return 1;
    }
    std::ifstream in(argv[1], std::ios::binary);
    if (!in) { std::cerr << "Failed to open " << argv[1] << std::endl; return 1; }
    int rank = 1;
    if (argc > 2) {
        rank = std::atoi(argv[2]);
        if (rank > 5) {
            std::cerr << "Failed to recognize the rank. Please type a number between 0 and 5, according to the following:\n"
                      << "0: species, 1: genus, 2: family, 3: order, 4:class, and 5: phylum." << std::endl;
            return 1;
        }
    }
    std::ofstream excluded("files_excluded.txt", std::ios::binary);
    size_t n_excluded = 0;
    std::string line;
    while (std::getline(in, line)) {
        const std::vector<std::string> f = fields(line);
        if (f.size() < 2) continue;                    // the reference indexes blindly; a blank line is skipped here
        if (f[1] == "-1") {
            if (n_excluded++ == 0) excluded << "The following files have been excluded from the targets definition" << std::endl;
            excluded << f[0] << std::endl;
            continue;
        }
        const size_t col = 2 + (size_t)rank;
        if (col < f.size() && f[col] != "UNKNOWN") std::cout << f[0] << "\t" << f[col] << std::endl;
    }
    return 0;
}
