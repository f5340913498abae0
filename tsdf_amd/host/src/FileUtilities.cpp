// reference: src/Utilities/FileUtilities.cpp
#include "FileUtilities.hpp"

#include <pwd.h>
#include <unistd.h>

#include <cctype>
#include <cstdlib>
#include <fstream>

#include <dirent.h>
#include <sys/stat.h>

#include <cstdio>
#include <fstream>

bool file_exists(const std::string &file_name, bool &is_directory) {
    struct stat st;
    is_directory = false;
    if (stat(file_name.c_str(), &st) != 0) return false;
    is_directory = S_ISDIR(st.st_mode);
    return true;
}

bool process_file_by_lines(const std::string &file_name, std::function<void(const std::string &)> processor) {
    std::ifstream f(file_name);
    if (!f.is_open()) {
        perror(("error while opening file " + file_name).c_str());
        return false;
    }
    std::string line;
    while (std::getline(f, line)) processor(line);
    if (f.bad()) perror(("error while reading file " + file_name).c_str());
    return true;
}

void files_in_directory(const std::string &directory, std::vector<std::string> &files, std::function<bool(const char *)> filter) {
    DIR *d = opendir(directory.c_str());
    if (!d) return;
    while (struct dirent *e = readdir(d))
        if (!filter || filter(e->d_name)) files.push_back(e->d_name);
    closedir(d);
}

// reference: src/Utilities/FileUtilities.cpp:29-83
bool match_file_name(const std::string &prefix, int num_digits, const std::string &suffix, const std::string &extension,
                     const std::string &test_string) {
    if (num_digits < 0) return false;
    const std::string tail = suffix + "." + extension;
    if (test_string.size() != prefix.size() + (size_t)num_digits + tail.size()) return false;
    if (test_string.compare(0, prefix.size(), prefix) != 0) return false;
    if (test_string.compare(prefix.size() + num_digits, tail.size(), tail) != 0) return false;
    for (int i = 0; i < num_digits; i++)
        if (!isdigit((unsigned char)test_string[prefix.size() + i])) return false;
    return true;
}

// reference: src/Utilities/FileUtilities.cpp:177-231 (scans backwards from the end; here the file is read forwards)
bool read_last_line(std::string file_name, std::string &text) {
    std::ifstream in(file_name);
    if (!in.is_open()) return false;
    bool found = false;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty()) {
            text = line;
            found = true;
        }
    }
    return found;
}

// reference: src/Utilities/FileUtilities.cpp:233-240
const char *get_home_directory() {
    const char *home = getenv("HOME");
    if (home) return home;
    const struct passwd *pw = getpwuid(getuid());
    return pw ? pw->pw_dir : nullptr;
}

// reference: src/Utilities/FileUtilities.cpp:246-259
const std::string path_to_file_on_desktop(const std::string &file_name) {
    const char *home = get_home_directory();
    return std::string(home ? home : "") + "/Desktop/" + file_name;
}
