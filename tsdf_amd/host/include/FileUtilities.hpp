// Small file helpers.  Same functions as the reference's src/include/FileUtilities.hpp (those the path uses).
#ifndef TSDF_AMD_HOST_FILE_UTILITIES_INCLUDED
#define TSDF_AMD_HOST_FILE_UTILITIES_INCLUDED

#include <functional>
#include <string>
#include <vector>

bool process_file_by_lines(const std::string &file_name, std::function<void(const std::string &)> processor);
bool file_exists(const std::string &file_name, bool &is_directory);
void files_in_directory(const std::string &directory, std::vector<std::string> &files, std::function<bool(const char *)> filter);
// "<prefix><num_digits digits><suffix>.<extension>" ?
bool match_file_name(const std::string &prefix, int num_digits, const std::string &suffix, const std::string &extension,
                     const std::string &test_string);
// last non-empty line of a text file
bool read_last_line(std::string file_name, std::string &text);
// $HOME, else the password database's entry
const char *get_home_directory();
// "<home>/Desktop/<file_name>"
const std::string path_to_file_on_desktop(const std::string &file_name);

#endif
