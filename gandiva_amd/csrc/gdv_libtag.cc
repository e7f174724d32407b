#include "gdv_libtag.h"

#include "gdv_runtime.h"

#include <cctype>
#include <set>

namespace gdv {

uint64_t Fnv1a64(const std::string& s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) {
    h ^= c;
    h *= 1099511628211ull;
  }
  return h;
}

namespace {

bool IsIdentStart(char c) { return std::isalpha(static_cast<unsigned char>(c)) || c == '_'; }
bool IsIdentChar(char c) { return std::isalnum(static_cast<unsigned char>(c)) || c == '_'; }

// comments -> one blank; string and character literals are kept as they are
std::string StripComments(const std::string& s) {
  std::string out;
  out.reserve(s.size());
  for (size_t i = 0; i < s.size();) {
    if (s[i] == '/' && i + 1 < s.size() && s[i + 1] == '/') {
      while (i < s.size() && s[i] != '\n') {
        // a line comment that ends in a backslash continues on the next line
        if (s[i] == '\\' && i + 1 < s.size() && s[i + 1] == '\n') i++;
        i++;
      }
      out.push_back(' ');
    } else if (s[i] == '/' && i + 1 < s.size() && s[i + 1] == '*') {
      i += 2;
      while (i + 1 < s.size() && !(s[i] == '*' && s[i + 1] == '/')) {
        if (s[i] == '\n') out.push_back('\n');  // keep line structure for the directive scanner
        i++;
      }
      i += 2;
      out.push_back(' ');
    } else if (s[i] == '"' || s[i] == '\'') {
      const char q = s[i];
      out.push_back(s[i++]);
      while (i < s.size() && s[i] != q) {
        if (s[i] == '\\' && i + 1 < s.size()) out.push_back(s[i++]);
        out.push_back(s[i++]);
      }
      if (i < s.size()) out.push_back(s[i++]);
    } else {
      out.push_back(s[i++]);
    }
  }
  return out;
}

std::string Collapse(const std::string& s) {
  std::string out;
  bool blank = true;
  for (char c : s) {
    if (std::isspace(static_cast<unsigned char>(c))) {
      if (!blank) out.push_back(' ');
      blank = true;
    } else {
      out.push_back(c);
      blank = false;
    }
  }
  while (!out.empty() && out.back() == ' ') out.pop_back();
  return out;
}

std::vector<std::string> Identifiers(const std::string& s) {
  std::set<std::string> seen;
  std::vector<std::string> out;
  for (size_t i = 0; i < s.size();) {
    if (s[i] == '"' || s[i] == '\'') {  // literals mention nothing
      const char q = s[i++];
      while (i < s.size() && s[i] != q) i += (s[i] == '\\') ? 2 : 1;
      i++;
    } else if (IsIdentStart(s[i])) {
      size_t j = i;
      while (j < s.size() && IsIdentChar(s[j])) j++;
      std::string id = s.substr(i, j - i);
      if (seen.insert(id).second) out.push_back(std::move(id));
      i = j;
    } else if (std::isdigit(static_cast<unsigned char>(s[i]))) {
      while (i < s.size() && (IsIdentChar(s[i]) || s[i] == '.')) i++;  // 0x80ull, 1e9 ...
    } else {
      i++;
    }
  }
  return out;
}

// `IDENT(...) IDENT(...) ...` with upper-case IDENTs and nothing else: instantiations of the
// type-family macros (GDV_NUMERIC_TYPES(GDV_RELOPS) ...), which carry no terminator
bool IsMacroInstantiation(const std::string& code) {
  size_t i = 0;
  int groups = 0;
  while (i < code.size()) {
    while (i < code.size() && code[i] == ' ') i++;
    if (i >= code.size()) break;
    size_t j = i;
    while (j < code.size() && (std::isupper(static_cast<unsigned char>(code[j])) ||
                               std::isdigit(static_cast<unsigned char>(code[j])) || code[j] == '_'))
      j++;
    if (j == i || j >= code.size() || code[j] != '(') return false;
    int depth = 0;
    for (; j < code.size(); j++) {
      if (code[j] == '(') depth++;
      else if (code[j] == ')' && --depth == 0) break;
      else if (code[j] == '{' || code[j] == '}' || code[j] == ';') return false;
    }
    if (j >= code.size()) return false;
    i = j + 1;
    groups++;
  }
  return groups > 0;
}

// the name a top-level declaration introduces ("" when it cannot be told)
std::string DeclaredName(const std::string& code) {
  size_t i = 0;
  auto skip_ws = [&] { while (i < code.size() && code[i] == ' ') i++; };
  auto word_at = [&](size_t p) {
    size_t q = p;
    while (q < code.size() && IsIdentChar(code[q])) q++;
    return code.substr(p, q - p);
  };
  skip_ws();
  while (word_at(i) == "template") {  // template <...> (possibly nested angle brackets)
    i += 8;
    skip_ws();
    if (i < code.size() && code[i] == '<') {
      int d = 0;
      for (; i < code.size(); i++) {
        if (code[i] == '<') d++;
        else if (code[i] == '>' && --d == 0) { i++; break; }
      }
    }
    skip_ws();
  }
  const std::string first = word_at(i);
  if (first == "struct" || first == "union" || first == "enum" || first == "class") {
    size_t p = i + first.size();
    while (p < code.size() && code[p] == ' ') p++;
    return word_at(p);
  }
  if (first == "typedef") {
    size_t e = code.rfind(';');
    if (e == std::string::npos) e = code.size();
    while (e > 0 && !IsIdentChar(code[e - 1])) e--;
    size_t b = e;
    while (b > 0 && IsIdentChar(code[b - 1])) b--;
    return code.substr(b, e - b);
  }
  // a function (name before the first '(' that does not belong to an attribute) or a variable
  // (name before '=' / '[' / ';')
  std::string last;
  for (size_t p = i; p < code.size();) {
    if (IsIdentStart(code[p])) {
      last = word_at(p);
      p += last.size();
    } else if (code[p] == '(') {
      if (last == "__attribute__" || last == "__launch_bounds__" || last == "alignas" || last == "__declspec") {
        int d = 0;
        for (; p < code.size(); p++) {
          if (code[p] == '(') d++;
          else if (code[p] == ')' && --d == 0) { p++; break; }
        }
        last.clear();
      } else {
        return last;
      }
    } else if (code[p] == '=' || code[p] == '[' || code[p] == ';' || code[p] == '{') {
      return last;
    } else {
      p++;
    }
  }
  return std::string();
}

}  // namespace

LibraryIndex::LibraryIndex(const std::string& header_source) {
  full_tag_ = std::to_string(Fnv1a64(header_source));
  const std::string src = StripComments(header_source);
  auto add = [&](const std::string& name, const std::string& code) {
    Item it;
    it.name = name;
    it.code = code;
    it.idents = Identifiers(code);
    if (!name.empty()) by_name_.emplace(name, items_.size());
    items_.push_back(std::move(it));
  };
  std::string cur;     // text of the item being collected
  int braces = 0, parens = 0;
  size_t pos = 0;
  while (pos < src.size()) {
    // one physical line at a time; directives are recognised at the start of a line outside braces
    size_t eol = src.find('\n', pos);
    if (eol == std::string::npos) eol = src.size();
    std::string line = src.substr(pos, eol - pos);
    pos = eol + 1;
    size_t first = line.find_first_not_of(" \t");
    if (braces == 0 && parens == 0 && first != std::string::npos && line[first] == '#' && Collapse(cur).empty()) {
      while (!line.empty() && line.back() == '\\' && pos < src.size()) {  // continuation lines
        line.pop_back();
        size_t e2 = src.find('\n', pos);
        if (e2 == std::string::npos) e2 = src.size();
        line += " " + src.substr(pos, e2 - pos);
        pos = e2 + 1;
      }
      const std::string code = Collapse(line);
      std::string name;
      size_t p = code.find_first_not_of("# ");
      if (code.compare(p, 6, "define") == 0) {
        p += 6;
        while (p < code.size() && code[p] == ' ') p++;
        size_t q = p;
        while (q < code.size() && IsIdentChar(code[q])) q++;
        name = code.substr(p, q - p);
      }
      add(name, code);  // #if / #ifndef / #else / #endif / #pragma / #undef: base
      cur.clear();
      continue;
    }
    for (size_t i = 0; i < line.size(); i++) {
      const char c = line[i];
      cur.push_back(c);
      if (c == '"' || c == '\'') {
        for (i++; i < line.size() && line[i] != c; i++) {
          cur.push_back(line[i]);
          if (line[i] == '\\' && i + 1 < line.size()) cur.push_back(line[++i]);
        }
        if (i < line.size()) cur.push_back(c);
        continue;
      }
      if (c == '(') parens++;
      else if (c == ')') parens--;
      else if (c == '{') braces++;
      else if (c == '}') {
        braces--;
        if (braces == 0 && parens == 0) {
          size_t k = i + 1;  // `};` closes a struct / an initialiser
          while (k < line.size() && (line[k] == ' ' || line[k] == '\t')) k++;
          if (k < line.size() && line[k] == ';') {
            cur.push_back(';');
            i = k;
          }
          const std::string code = Collapse(cur);
          add(DeclaredName(code), code);
          cur.clear();
        }
      } else if (c == ';' && braces == 0 && parens == 0) {
        const std::string code = Collapse(cur);
        add(DeclaredName(code), code);
        cur.clear();
      }
    }
    cur.push_back('\n');
    if (braces == 0 && parens == 0) {
      const std::string code = Collapse(cur);
      if (!code.empty() && IsMacroInstantiation(code)) {
        add(std::string(), code);
        cur.clear();
      }
    }
  }
  const std::string rest = Collapse(cur);
  if (!rest.empty()) add(std::string(), rest);
}

std::vector<bool> LibraryIndex::Reach(const std::string& kernel_text) const {
  std::vector<bool> in(items_.size(), false);
  std::vector<size_t> work;
  auto visit = [&](const std::vector<std::string>& ids) {
    for (auto& id : ids) {
      auto range = by_name_.equal_range(id);
      for (auto it = range.first; it != range.second; ++it) {
        if (!in[it->second]) {
          in[it->second] = true;
          work.push_back(it->second);
        }
      }
    }
  };
  for (size_t k = 0; k < items_.size(); k++) {
    if (items_[k].name.empty()) {
      in[k] = true;
      work.push_back(k);
    }
  }
  visit(Identifiers(StripComments(kernel_text)));
  while (!work.empty()) {
    const size_t k = work.back();
    work.pop_back();
    visit(items_[k].idents);
  }
  return in;
}

std::string LibraryIndex::TagFor(const std::string& kernel_text) const {
  const std::vector<bool> in = Reach(kernel_text);
  uint64_t h = 1469598103934665603ull;
  for (size_t k = 0; k < items_.size(); k++) {
    if (!in[k]) continue;
    for (unsigned char c : items_[k].code) {
      h ^= c;
      h *= 1099511628211ull;
    }
    h ^= 0xff;  // item separator
    h *= 1099511628211ull;
  }
  return std::to_string(h);
}

std::vector<std::string> LibraryIndex::ReachedFrom(const std::string& kernel_text) const {
  const std::vector<bool> in = Reach(kernel_text);
  std::vector<std::string> out;
  for (size_t k = 0; k < items_.size(); k++)
    if (in[k]) out.push_back(items_[k].name.empty() ? "<base> " + items_[k].code.substr(0, 60) : items_[k].name);
  return out;
}

const LibraryIndex& LibraryIndex::Embedded() {
  static const LibraryIndex idx{std::string(gdv_device_lib_src)};
  return idx;
}

}  // namespace gdv
