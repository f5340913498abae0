// reference: src/Utilities/FileUtilities.cpp
#include "FileUtilities.hpp"

#include <pwd.h>
#include <unistd.h>

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>

#include <dirent.h>
#include <sys/stat.h>

#include <cstdio>
#include <fstream>

// reference: src/Utilities/FileUtilities.cpp:117-131 (is_directory is written for plain files and directories only)
bool file_exists(const std::string &file_name, bool &is_directory) {
    struct stat st;
    if (stat(file_name.c_str(), &st) != 0) return false;
    if (S_ISREG(st.st_mode)) is_directory = false;
    else if (S_ISDIR(st.st_mode)) is_directory = true;
    return true;
}

// reference: src/Utilities/FileUtilities.cpp:85-110 (a file that does not open is reported on stderr and the call still returns true)
bool process_file_by_lines(const std::string &file_name, std::function<void(const std::string &)> processor) {
    std::ifstream f(file_name);
    if (!f.is_open()) perror(("error while opening file " + file_name).c_str());
    std::string line;
    while (std::getline(f, line)) processor(line);
    if (f.bad()) perror(("error while reading file " + file_name).c_str());
    return true;
}

// reference: src/Utilities/FileUtilities.cpp:140-160 (the names in the directory's own order)
void files_in_directory(const std::string &directory, std::vector<std::string> &files, std::function<bool(const char *)> filter) {
    DIR *d = opendir(directory.c_str());
    if (!d) {
        std::cerr << "Problem reading directory " << directory << std::endl;
        return;
    }
    while (struct dirent *e = readdir(d))
        if (!filter || filter(e->d_name)) files.push_back(e->d_name);
    closedir(d);
}

// reference: src/Utilities/FileUtilities.cpp:29-83: <prefix><digits><suffix><any one character><extension> -- the character in
// front of the extension is counted, not compared
bool match_file_name(const std::string &prefix, int num_digits, const std::string &suffix, const std::string &extension,
                     const std::string &test_string) {
    if (num_digits < 0) return false;
    if (test_string.size() != prefix.size() + (size_t)num_digits + suffix.size() + 1 + extension.size()) return false;
    if (test_string.compare(0, prefix.size(), prefix) != 0) return false;
    if (test_string.compare(prefix.size() + num_digits, suffix.size(), suffix) != 0) return false;
    if (test_string.compare(test_string.size() - extension.size(), extension.size(), extension) != 0) return false;
    for (int i = 0; i < num_digits; i++)
        if (!isdigit((unsigned char)test_string[prefix.size() + i])) return false;
    return true;
}

// reference: src/Utilities/FileUtilities.cpp:177-231.  The reference walks back from the end of the file to the last line that holds
// a character (a carriage return is one) and reads it forwards again; when that line is the file's FIRST its walk has left the
// stream failed, the read returns nothing and the call still reports success with `text` untouched.  Kept: here the file is read
// once, forwards.
bool read_last_line(std::string file_name, std::string &text) {
    std::ifstream in(file_name, std::ios::binary);
    if (!in.is_open()) {
        std::cerr << "Couldn't read file " << file_name << std::endl;
        return false;
    }
    const std::string all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t end = all.size();
    while (end > 0) {
        size_t begin = all.rfind('\n', end - 1);
        begin = begin == std::string::npos ? 0 : begin + 1;
        if (begin < end) {
            if (begin > 0) text = all.substr(begin, end - begin);
            return true;
        }
        end = begin - 1;   // (an empty line: begin == end, the newline in front of it is at begin - 1)
    }
    return false;
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
