#include "yaml_lite.hpp"

#include <fstream>
#include <sstream>

namespace eagle_mpc {
namespace yaml_lite {

static const Node kUndefined;

const Node& Node::operator[](const std::string& key) const {
  if (type == Map)
    for (const auto& kv : map)
      if (kv.first == key) return kv.second;
  return kUndefined;
}

namespace {

struct Line {
  int indent;
  std::string text;  // comment-stripped, right-trimmed, without the indentation
  int number;
};

std::string rtrim(const std::string& s) {
  size_t e = s.size();
  while (e > 0 && (s[e - 1] == ' ' || s[e - 1] == '\t' || s[e - 1] == '\r')) --e;
  return s.substr(0, e);
}
std::string trim(const std::string& s) {
  size_t b = 0;
  while (b < s.size() && (s[b] == ' ' || s[b] == '\t')) ++b;
  return rtrim(s.substr(b));
}

// remove a trailing comment: '#' at line start or preceded by whitespace, outside quotes
std::string strip_comment(const std::string& s) {
  bool in_d = false, in_s = false;
  for (size_t i = 0; i < s.size(); ++i) {
    const char c = s[i];
    if (c == '"' && !in_s) in_d = !in_d;
    if (c == '\'' && !in_d) in_s = !in_s;
    if (c == '#' && !in_d && !in_s && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
  }
  return s;
}

std::string unquote(const std::string& s) {
  if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
    return s.substr(1, s.size() - 2);
  return s;
}

Node make_scalar(const std::string& raw) {
  Node n;
  const std::string t = trim(raw);
  if (t.empty() || t == "~" || t == "null") {
    n.type = Node::Null;
    return n;
  }
  n.type = Node::Scalar;
  n.scalar = unquote(t);
  return n;
}

struct Parser {
  std::vector<Line> lines;
  size_t pos = 0;

  [[noreturn]] void fail(const std::string& msg, int line) {
    throw std::runtime_error("yaml_lite: " + msg + " (line " + std::to_string(line) + ")");
  }

  // flow sequence starting in `first` (which begins with '['); may continue on following lines
  Node parse_flow(std::string text, int line_no) {
    int depth = 0;
    auto balance = [&](const std::string& s) {
      for (char c : s) {
        if (c == '[') ++depth;
        if (c == ']') --depth;
      }
    };
    balance(text);
    while (depth > 0) {
      if (pos >= lines.size()) fail("unterminated flow sequence", line_no);
      text += " " + lines[pos].text;
      balance(lines[pos].text);
      ++pos;
    }
    return parse_flow_text(trim(text), line_no);
  }
  Node parse_flow_text(const std::string& t, int line_no) {
    if (t.size() < 2 || t.front() != '[' || t.back() != ']') fail("malformed flow sequence '" + t + "'", line_no);
    Node n;
    n.type = Node::Sequence;
    const std::string inner = t.substr(1, t.size() - 2);
    int depth = 0;
    std::string cur;
    auto flush = [&]() {
      const std::string item = trim(cur);
      if (!item.empty()) {
        if (item.front() == '[')
          n.seq.push_back(parse_flow_text(item, line_no));
        else
          n.seq.push_back(make_scalar(item));
      }
      cur.clear();
    };
    for (char c : inner) {
      if (c == '[') ++depth;
      if (c == ']') --depth;
      if (c == ',' && depth == 0)
        flush();
      else
        cur += c;
    }
    flush();
    return n;
  }

  // value that follows "key:" (or "- ") on the same line
  Node parse_inline_value(const std::string& rest, int line_no) {
    const std::string t = trim(rest);
    if (!t.empty() && t.front() == '[') return parse_flow(t, line_no);
    return make_scalar(t);
  }

  // find "key:" split; returns npos when the text is not a mapping entry
  static size_t find_colon(const std::string& s) {
    bool in_d = false, in_s = false;
    for (size_t i = 0; i < s.size(); ++i) {
      const char c = s[i];
      if (c == '"' && !in_s) in_d = !in_d;
      if (c == '\'' && !in_d) in_s = !in_s;
      if (c == '[' && !in_d && !in_s) return std::string::npos;
      if (c == ':' && !in_d && !in_s && (i + 1 == s.size() || s[i + 1] == ' ')) return i;
    }
    return std::string::npos;
  }

  Node parse_block(int indent) {
    if (pos >= lines.size()) {
      Node n;
      n.type = Node::Null;
      return n;
    }
    const Line& first = lines[pos];
    if (first.text.compare(0, 1, "[") == 0) {
      std::string t = first.text;
      const int ln = first.number;
      ++pos;
      return parse_flow(t, ln);
    }
    if (first.text == "-" || first.text.compare(0, 2, "- ") == 0) return parse_sequence(indent);
    if (find_colon(first.text) != std::string::npos) return parse_map(indent);
    // bare scalar block
    Node n = make_scalar(first.text);
    ++pos;
    return n;
  }

  Node parse_sequence(int indent) {
    Node n;
    n.type = Node::Sequence;
    while (pos < lines.size() && lines[pos].indent == indent &&
           (lines[pos].text == "-" || lines[pos].text.compare(0, 2, "- ") == 0)) {
      Line& l = lines[pos];
      std::string rest = l.text.size() > 1 ? l.text.substr(2) : "";
      size_t lead = 0;
      while (lead < rest.size() && rest[lead] == ' ') ++lead;
      rest = rest.substr(lead);
      const int item_indent = indent + 2 + (int)lead;
      if (rest.empty()) {
        ++pos;
        if (pos < lines.size() && lines[pos].indent > indent)
          n.seq.push_back(parse_block(lines[pos].indent));
        else {
          Node nul;
          nul.type = Node::Null;
          n.seq.push_back(nul);
        }
      } else if (find_colon(rest) != std::string::npos) {
        // "- key: value" opens a mapping whose entries sit at item_indent
        l.indent = item_indent;
        l.text = rest;
        n.seq.push_back(parse_map(item_indent));
      } else {
        const int ln = l.number;
        ++pos;
        n.seq.push_back(parse_inline_value(rest, ln));
      }
    }
    return n;
  }

  Node parse_map(int indent) {
    Node n;
    n.type = Node::Map;
    while (pos < lines.size() && lines[pos].indent == indent) {
      const Line l = lines[pos];
      if (l.text == "-" || l.text.compare(0, 2, "- ") == 0) break;
      const size_t c = find_colon(l.text);
      if (c == std::string::npos) fail("expected 'key: value', got '" + l.text + "'", l.number);
      const std::string key = unquote(trim(l.text.substr(0, c)));
      const std::string rest = trim(l.text.substr(c + 1));
      ++pos;
      Node value;
      if (!rest.empty()) {
        value = parse_inline_value(rest, l.number);
      } else if (pos < lines.size() && lines[pos].indent > indent) {
        value = parse_block(lines[pos].indent);
      } else if (pos < lines.size() && lines[pos].indent == indent &&
                 (lines[pos].text == "-" || lines[pos].text.compare(0, 2, "- ") == 0)) {
        value = parse_sequence(indent);  // sequence at the indentation of its key
      } else {
        value.type = Node::Null;
      }
      n.map.emplace_back(key, value);
    }
    if (pos < lines.size() && lines[pos].indent > indent)
      fail("unexpected indentation at '" + lines[pos].text + "'", lines[pos].number);
    return n;
  }
};

}  // namespace

Node load_string(const std::string& text) {
  Parser p;
  std::istringstream ss(text);
  std::string raw;
  int number = 0;
  while (std::getline(ss, raw)) {
    ++number;
    std::string s = rtrim(strip_comment(raw));
    size_t b = 0;
    while (b < s.size() && s[b] == ' ') ++b;
    if (b < s.size() && s[b] == '\t') throw std::runtime_error("yaml_lite: tab indentation (line " + std::to_string(number) + ")");
    if (b == s.size()) continue;
    if (s.compare(b, 3, "---") == 0) continue;
    p.lines.push_back({(int)b, s.substr(b), number});
  }
  if (p.lines.empty()) {
    Node n;
    n.type = Node::Null;
    return n;
  }
  Node root = p.parse_block(p.lines[0].indent);
  if (p.pos < p.lines.size())
    throw std::runtime_error("yaml_lite: trailing content at line " + std::to_string(p.lines[p.pos].number));
  return root;
}

Node load_file(const std::string& path) {
  std::ifstream f(path);
  if (!f.good()) throw std::runtime_error("Couldn't load file: " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return load_string(ss.str());
}

}  // namespace yaml_lite
}  // namespace eagle_mpc
