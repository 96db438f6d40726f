// .ini reader with the semantics of the reference's configuration files.
//
// Reference behaviour: inih (inih/ini.c:61-165) + INIReader (inih/cpp/INIReader.cpp):
//   - lines are at most 199 characters; "[section]"; "name = value" or "name: value";
//   - a ';' starts a comment at the start of a line, or inside a line when preceded by white space;
//     on top of that every typed getter cuts the value at its first ';' (INIReader.cpp:36-118), which is
//     what makes "TiltSet=1,2,4;comment" work;
//   - a non-blank line that starts with white space continues the previous value ("\n" joined);
//   - a repeated key appends to the earlier value with "\n" (INIReader::ValueHandler);
//   - section and key names are case-insensitive (MakeKey lower-cases both).
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace modscli {

class IniReader {
 public:
  explicit IniReader(const std::string &filename) { error_ = parse(filename); }
  int ParseError() const { return error_; }   // 0 ok, -1 cannot open, > 0 first bad line

  std::string Get(const std::string &section, const std::string &name, const std::string &def) const {
    auto it = values_.find(key(section, name));
    return it == values_.end() ? def : it->second;
  }
  bool Has(const std::string &section, const std::string &name) const { return values_.count(key(section, name)) != 0; }
  std::string GetString(const std::string &section, const std::string &name, const std::string &def) const {
    (void)def;                                  // the reference ignores the default here (INIReader.cpp:36-45)
    return cut(Get(section, name, ""));
  }
  long GetInteger(const std::string &section, const std::string &name, long def) const {
    const std::string v = cut(Get(section, name, ""));
    char *end;
    const long n = strtol(v.c_str(), &end, 0);
    return end > v.c_str() ? n : def;
  }
  double GetDouble(const std::string &section, const std::string &name, double def) const {
    const std::string v = cut(Get(section, name, ""));
    char *end;
    const double n = strtod(v.c_str(), &end);
    return end > v.c_str() ? n : def;
  }
  bool GetBoolean(const std::string &section, const std::string &name, bool def) const {
    std::string v = Get(section, name, "");
    std::transform(v.begin(), v.end(), v.begin(), ::tolower);
    v = cut(v);
    if (v == "true" || v == "yes" || v == "on" || v == "1") return true;
    if (v == "false" || v == "no" || v == "off" || v == "0") return false;
    return def;
  }
  // comma separated lists; a value without a comma is a one-element list (also when it is empty:
  // strtod("") = 0, INIReader.cpp:97-103)
  std::vector<double> GetDoubleVector(const std::string &section, const std::string &name) const {
    std::vector<double> out;
    for (const std::string &s : split(cut(Get(section, name, "")))) out.push_back(strtod(s.c_str(), nullptr));
    return out;
  }
  std::vector<std::string> GetStringVector(const std::string &section, const std::string &name) const { return split(cut(Get(section, name, ""))); }

 private:
  std::map<std::string, std::string> values_;
  int error_ = 0;

  static std::string key(const std::string &section, const std::string &name) {
    std::string k = section + "." + name;
    std::transform(k.begin(), k.end(), k.begin(), ::tolower);
    return k;
  }
  static std::string cut(const std::string &v) {
    const size_t p = v.find(';');
    return p == std::string::npos ? v : v.substr(0, p);
  }
  static std::vector<std::string> split(const std::string &v) {
    std::vector<std::string> out;
    size_t prev = 0, found = v.find(',');
    if (found == std::string::npos) { out.push_back(v); return out; }
    while (found != std::string::npos) {
      out.push_back(v.substr(prev, found - prev));
      prev = found + 1;
      found = v.find(',', prev);
    }
    out.push_back(v.substr(prev));
    return out;
  }
  static char *rstrip(char *s) {
    char *p = s + strlen(s);
    while (p > s && isspace((unsigned char)(*--p))) *p = '\0';
    return s;
  }
  static char *lskip(char *s) {
    while (*s && isspace((unsigned char)(*s))) s++;
    return s;
  }
  // first occurrence of c, or of a ';' that follows white space, or the terminating NUL
  static char *find_char_or_comment(char *s, char c) {
    int was_space = 0;
    while (*s && *s != c && !(was_space && *s == ';')) {
      was_space = isspace((unsigned char)(*s));
      s++;
    }
    return s;
  }
  void store(const std::string &section, const std::string &name, const std::string &value) {
    const std::string k = key(section, name);
    if (values_.count(k) && !values_[k].empty()) values_[k] += "\n";
    values_[k] += value;
  }
  int parse(const std::string &filename) {
    FILE *f = fopen(filename.c_str(), "r");
    if (!f) return -1;
    char line[200];
    std::string section, prev_name;
    int lineno = 0, error = 0;
    while (fgets(line, sizeof(line), f) != nullptr) {
      lineno++;
      char *start = line;
      if (lineno == 1 && (unsigned char)start[0] == 0xEF && (unsigned char)start[1] == 0xBB && (unsigned char)start[2] == 0xBF) start += 3;
      start = lskip(rstrip(start));
      if (*start == ';' || *start == '#') continue;
      if (!prev_name.empty() && *start && start > line) { store(section, prev_name, start); continue; }
      if (*start == '[') {
        char *end = find_char_or_comment(start + 1, ']');
        if (*end == ']') {
          *end = '\0';
          section = std::string(start + 1).substr(0, 49);
          prev_name.clear();
        } else if (!error) error = lineno;
      } else if (*start) {
        char *end = find_char_or_comment(start, '=');
        if (*end != '=') end = find_char_or_comment(start, ':');
        if (*end == '=' || *end == ':') {
          *end = '\0';
          char *name = rstrip(start);
          char *value = lskip(end + 1);
          end = find_char_or_comment(value, '\0');
          if (*end == ';') *end = '\0';
          rstrip(value);
          prev_name = std::string(name).substr(0, 49);
          store(section, prev_name, value);
        } else if (!error) error = lineno;
      }
    }
    fclose(f);
    return error;
  }
};

}  // namespace modscli
