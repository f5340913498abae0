// Small file helpers.  Same functions as the reference's src/include/FileUtilities.hpp (those the path uses).
#ifndef TSDF_AMD_HOST_FILE_UTILITIES_INCLUDED
#define TSDF_AMD_HOST_FILE_UTILITIES_INCLUDED

#include <functional>
#include <string>
#include <vector>

bool process_file_by_lines(const std::string &file_name, std::function<void(const std::string &)> processor);
bool file_exists(const std::string &file_name, bool &is_directory);
void files_in_directory(const std::string &directory, std::vector<std::string> &files, std::function<bool(const char *)> filter);

#endif
