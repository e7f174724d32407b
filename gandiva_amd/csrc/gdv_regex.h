// Regular expressions for regexp_like / regexp_matches (round 5): the pattern — a literal of the expression, as for the
// reference's holders, which compile it with RE2 once per expression [regexp_matches_holder.cc / like_holder.cc, as
// recalled] — is compiled HERE, at Make time, into a position automaton (Glushkov: one state per character position of the
// pattern, no epsilon moves) of at most 63 positions, laid out as a table the device function gdv_regex_search walks with
// one 64-bit state set per row:
//     uint64 first, last, flags (1 nullable, 2 anchored at the start, 4 anchored at the end), follow[64], match[256]
// (match[b] = the positions that accept byte b).  "Does the text contain a match" is all the two functions ask, so
// greedy / lazy / leftmost-first make no difference and the automaton's answer is RE2's.
//
// Syntax taken: literals, '.', classes [a-z0-9_] and [^...] (ASCII members), \d \D \w \W \s \S, \t \n \r \f \v \xHH and
// escaped punctuation, groups ( ) and (?: ), alternation |, quantifiers * + ? {m} {m,} {m,n} (and their lazy forms), '^'
// as the pattern's first and '$' as its last character, '(?i)' in front of everything (ASCII letters in either case), non-ASCII
// characters as members of a positive class.  '.' and negated classes consume a whole UTF-8 character (a lead
// byte and its continuation bytes); '.' does not match a newline (RE2's default).  Anything else — back-references, \b,
// look-around, other flags, anchors inside the pattern, non-ASCII range ends or negated classes with non-ASCII members, (?i) next to
// non-ASCII characters, more than 63 positions — is
// refused with a message, not guessed.
#pragma once
#include <string>

#include "gdv_types.h"

namespace gdv {

// table = the bytes described above (2584 of them); Invalid / CodeGenError with the reason otherwise
Status CompileRegex(const std::string& pattern, std::string* table);

// to_date's SQL pattern -> one byte per strptime directive, the program gdv_parse_date interprets (gdv_regex.cc)
Status CompileDateFormat(const std::string& pattern, std::string* ops);

}  // namespace gdv
