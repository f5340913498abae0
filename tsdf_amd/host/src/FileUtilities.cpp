// reference: src/Utilities/FileUtilities.cpp
#include "FileUtilities.hpp"

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
