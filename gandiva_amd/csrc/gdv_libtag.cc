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

// ---- function-like macros of the header: expanding the type-family instantiations -------------
// `GDV_NUMERIC_TYPES(GDV_RELOPS)` carries no name, but what it EXPANDS to does: ten times six
// functions, each with a name of its own.  Rounds 3-4 hashed such lines (and, through the
// identifiers they mention, the macros behind them) into every kernel; a new `GDV_DATE_TRUNC(T)`
// family then renamed the float64 projection and orphaned its profiles.  The instantiations are
// expanded here instead — parameters substituted, `##` pasted, the result rescanned at brace
// depth 0 (X-macro lists hand macro names on as arguments) — and split into ordinary named items.
// Nothing inside a function body is expanded: a macro used there is an identifier like any other
// and reached by name.  The header uses neither `#param` nor variadic macros; a line that does
// not expand cleanly stays a base item as before.
struct FnMacro {
  std::vector<std::string> params;
  std::string body;
};

std::vector<std::string> SplitArgs(const std::string& s) {
  std::vector<std::string> out;
  std::string cur;
  int depth = 0;
  for (size_t i = 0; i < s.size(); i++) {
    const char c = s[i];
    if (c == '"' || c == '\'') {
      cur.push_back(c);
      for (i++; i < s.size() && s[i] != c; i++) {
        cur.push_back(s[i]);
        if (s[i] == '\\' && i + 1 < s.size()) cur.push_back(s[++i]);
      }
      if (i < s.size()) cur.push_back(c);
      continue;
    }
    if (c == '(' || c == '[' || c == '{') depth++;
    else if (c == ')' || c == ']' || c == '}') depth--;
    if (c == ',' && depth == 0) {
      out.push_back(Collapse(cur));
      cur.clear();
    } else {
      cur.push_back(c);
    }
  }
  out.push_back(Collapse(cur));
  return out;
}

// `#define NAME(a, b) body` (already collapsed to one line) -> table entry; false for object-like macros
bool ParseFnMacro(const std::string& code, std::string* name, FnMacro* m) {
  size_t p = code.find("define");
  if (p == std::string::npos) return false;
  p += 6;
  while (p < code.size() && code[p] == ' ') p++;
  size_t q = p;
  while (q < code.size() && IsIdentChar(code[q])) q++;
  if (q == p || q >= code.size() || code[q] != '(') return false;  // a blank before '(' = object-like
  *name = code.substr(p, q - p);
  size_t close = code.find(')', q);
  if (close == std::string::npos) return false;
  const std::string plist = code.substr(q + 1, close - q - 1);
  if (plist.find("...") != std::string::npos) return false;
  m->params.clear();
  if (!Collapse(plist).empty())
    for (auto& a : SplitArgs(plist)) m->params.push_back(a);
  m->body = code.substr(close + 1);
  return true;
}

bool Substitute(const FnMacro& m, const std::vector<std::string>& args, std::string* out) {
  if (args.size() != m.params.size()) return false;
  std::string r;
  const std::string& b = m.body;
  for (size_t i = 0; i < b.size();) {
    if (b[i] == '"' || b[i] == '\'') {
      const char q = b[i];
      r.push_back(b[i++]);
      while (i < b.size() && b[i] != q) {
        if (b[i] == '\\' && i + 1 < b.size()) r.push_back(b[i++]);
        r.push_back(b[i++]);
      }
      if (i < b.size()) r.push_back(b[i++]);
    } else if (IsIdentStart(b[i])) {
      size_t j = i;
      while (j < b.size() && IsIdentChar(b[j])) j++;
      const std::string id = b.substr(i, j - i);
      size_t k = 0;
      for (; k < m.params.size(); k++)
        if (m.params[k] == id) break;
      r += k < m.params.size() ? args[k] : id;
      i = j;
    } else if (std::isdigit(static_cast<unsigned char>(b[i]))) {
      while (i < b.size() && (IsIdentChar(b[i]) || b[i] == '.')) r.push_back(b[i++]);
    } else if (b[i] == '#' && i + 1 < b.size() && b[i + 1] == '#') {
      while (!r.empty() && r.back() == ' ') r.pop_back();
      i += 2;
      while (i < b.size() && b[i] == ' ') i++;
    } else if (b[i] == '#') {
      return false;  // stringification: not used by the header, not modelled
    } else {
      r.push_back(b[i++]);
    }
  }
  *out = r;
  return true;
}

// expands macro invocations found at brace depth 0 of `in` (recursively); false = leave the line alone
bool ExpandTopLevel(const std::map<std::string, FnMacro>& macros, const std::string& in, std::string* out, int guard) {
  if (guard > 16) return false;
  int braces = 0;
  for (size_t i = 0; i < in.size();) {
    const char c = in[i];
    if (c == '"' || c == '\'') {
      out->push_back(in[i++]);
      while (i < in.size() && in[i] != c) {
        if (in[i] == '\\' && i + 1 < in.size()) out->push_back(in[i++]);
        out->push_back(in[i++]);
      }
      if (i < in.size()) out->push_back(in[i++]);
      continue;
    }
    if (c == '{') braces++;
    if (c == '}') braces--;
    if (braces == 0 && IsIdentStart(c) && (i == 0 || !IsIdentChar(in[i - 1]))) {
      size_t j = i;
      while (j < in.size() && IsIdentChar(in[j])) j++;
      const std::string id = in.substr(i, j - i);
      auto it = macros.find(id);
      size_t k = j;
      while (k < in.size() && in[k] == ' ') k++;
      if (it != macros.end() && k < in.size() && in[k] == '(') {
        int depth = 0;
        size_t e = k;
        for (; e < in.size(); e++) {
          if (in[e] == '(') depth++;
          else if (in[e] == ')' && --depth == 0) break;
        }
        if (e >= in.size()) return false;
        const std::string inner = in.substr(k + 1, e - k - 1);
        std::vector<std::string> args;
        if (!(it->second.params.empty() && Collapse(inner).empty())) args = SplitArgs(inner);
        std::string once, deep;
        if (!Substitute(it->second, args, &once)) return false;
        if (!ExpandTopLevel(macros, once, &deep, guard + 1)) return false;
        *out += deep;
        out->push_back(' ');
        i = e + 1;
        continue;
      }
      *out += id;
      i = j;
      continue;
    }
    out->push_back(in[i++]);
  }
  return true;
}

// top-level declarations of a text without directives (the expansion of a family)
std::vector<std::string> SplitTopLevel(const std::string& text) {
  std::vector<std::string> out;
  std::string cur;
  int braces = 0, parens = 0;
  for (size_t i = 0; i < text.size(); i++) {
    const char c = text[i];
    cur.push_back(c);
    if (c == '"' || c == '\'') {
      for (i++; i < text.size() && text[i] != c; i++) {
        cur.push_back(text[i]);
        if (text[i] == '\\' && i + 1 < text.size()) cur.push_back(text[++i]);
      }
      if (i < text.size()) cur.push_back(c);
      continue;
    }
    if (c == '(') parens++;
    else if (c == ')') parens--;
    else if (c == '{') braces++;
    else if (c == '}') {
      braces--;
      if (braces == 0 && parens == 0) {
        size_t k = i + 1;
        while (k < text.size() && text[k] == ' ') k++;
        if (k < text.size() && text[k] == ';') {
          cur.push_back(';');
          i = k;
        }
        out.push_back(Collapse(cur));
        cur.clear();
      }
    } else if (c == ';' && braces == 0 && parens == 0) {
      out.push_back(Collapse(cur));
      cur.clear();
    }
  }
  const std::string rest = Collapse(cur);
  if (!rest.empty()) out.push_back(rest);
  return out;
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
  std::map<std::string, FnMacro> fn_macros;  // the header's function-like macros, as met so far
  // an instantiation line: its expansion as named items; a base item when it does not expand
  auto add_instantiation = [&](const std::string& code) {
    std::string expanded;
    if (ExpandTopLevel(fn_macros, code, &expanded, 0)) {
      const std::vector<std::string> decls = SplitTopLevel(Collapse(expanded));
      bool all_named = !decls.empty();
      for (auto& d : decls) all_named = all_named && !DeclaredName(d).empty() && d.find('{') != std::string::npos;
      if (all_named) {
        for (auto& d : decls) add(DeclaredName(d), d);
        return;
      }
    }
    add(std::string(), code);
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
        std::string mname;
        FnMacro fm;
        if (ParseFnMacro(code, &mname, &fm)) fn_macros[mname] = std::move(fm);
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
        add_instantiation(code);
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
