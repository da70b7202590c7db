#include "params.hpp"

#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <stack>

namespace eagle_mpc {

// ---------------------------------------------------------------------------------------------------
// data directories
// ---------------------------------------------------------------------------------------------------
static std::string g_yaml_dir, g_robot_dir;
void set_yaml_dir(const std::string& dir) { g_yaml_dir = dir; }
void set_robot_data_dir(const std::string& dir) { g_robot_dir = dir; }
const std::string& yaml_dir() {
  if (g_yaml_dir.empty()) {
    const char* e = std::getenv("EAGLE_MPC_YAML_DIR");
    if (e) g_yaml_dir = e;
  }
  return g_yaml_dir;
}
const std::string& robot_data_dir() {
  if (g_robot_dir.empty()) {
    const char* e = std::getenv("EAGLE_MPC_ROBOT_DATA_DIR");
    if (e) g_robot_dir = e;
  }
  return g_robot_dir;
}
std::string getYamlPath(const std::string& p) { return p.find("/", 0) == 0 ? p : yaml_dir() + "/" + p; }
std::string getUrdfPath(const std::string& p) { return p.find("/", 0) == 0 ? p : robot_data_dir() + "/" + p; }

// ---------------------------------------------------------------------------------------------------
// converters
// ---------------------------------------------------------------------------------------------------
std::vector<std::string> parseList(const std::string& val) {
  std::stack<char> limiters;
  std::stack<std::string> word_stack;
  std::string current_word;
  std::vector<std::string> words;
  for (const char current : val) {
    if (current == '[' || current == '{') {
      limiters.push(current);
      word_stack.push(current_word);
      current_word = "";
    } else if (current == ']' || current == '}') {
      const char open = current == ']' ? '[' : '{';
      if (limiters.empty() || limiters.top() != open) throw std::runtime_error("Unmatched delimiter");
      if (limiters.size() > 1) {
        if (word_stack.empty()) word_stack.push("");
        current_word = word_stack.top() + std::string(1, open) + current_word + std::string(1, current);
        word_stack.pop();
      } else if (current == '}' || current_word != "") {
        words.push_back(current_word);
      }
      limiters.pop();
    } else if (current == ',') {
      if (limiters.size() == 1 && current_word != "") {
        words.push_back(current_word);
        current_word = "";
      } else if (limiters.size() > 1) {
        current_word += current;
      }
    } else {
      if (limiters.empty()) throw std::runtime_error("Found non-delimited text");
      current_word += current;
    }
  }
  if (!limiters.empty()) throw std::runtime_error("Unclosed delimiter [] or {}");
  return words;
}

static bool is_list_string(const std::string& v) { return v.size() >= 2 && v.front() == '[' && v.back() == ']'; }

template <>
int converter<int>::convert(const std::string& val) {
  return std::stoi(val);
}
template <>
double converter<double>::convert(const std::string& val) {
  return std::stod(val);
}
template <>
bool converter<bool>::convert(const std::string& val) {
  if (val == "true") return true;
  if (val == "false") return false;
  throw std::runtime_error("Invalid conversion to bool (Must be either \"true\" or \"false\"). String provided: " + val);
}
template <>
std::string converter<std::string>::convert(const std::string& val) {
  return val;
}

// element grammar of numeric vectors: -?[0-9]*(\.[0-9]+)?   (src/utils/converter_utils.cpp:39-40)
static bool is_plain_number(const std::string& s) {
  size_t i = 0;
  if (i < s.size() && s[i] == '-') ++i;
  while (i < s.size() && s[i] >= '0' && s[i] <= '9') ++i;
  if (i < s.size() && s[i] == '.') {
    ++i;
    const size_t d = i;
    while (i < s.size() && s[i] >= '0' && s[i] <= '9') ++i;
    if (i == d) return false;
  }
  return i == s.size();
}

template <>
VectorXd converter<VectorXd>::convert(const std::string& val) {
  const std::string err =
      "Invalid string representation of a Matrix. Correct format is [([num,num],)?(num(,num)*)?]. String provided: " + val;
  if (!is_list_string(val)) throw std::runtime_error(err);
  VectorXd out;
  std::string cur;
  const std::string inner = val.substr(1, val.size() - 2);
  auto flush = [&]() {
    if (!is_plain_number(cur) || cur.empty() || cur == "-") throw std::runtime_error(err);
    out.push_back(std::stod(cur));
    cur.clear();
  };
  if (inner.empty()) return out;
  for (char c : inner) {
    if (c == ',')
      flush();
    else
      cur += c;
  }
  flush();
  return out;
}

template <>
std::vector<std::string> converter<std::vector<std::string>>::convert(const std::string& val) {
  if (!is_list_string(val))
    throw std::runtime_error(
        "Invalid string format representing a list-like structure. Correct format is [(value)?(,value)*]. String "
        "provided: " +
        val);
  return parseList(val);
}

static std::pair<std::string, std::string> split_pair(const std::string& val) {
  // {identifier:value}
  const size_t c = val.find(':');
  if (val.size() < 5 || val.front() != '{' || val.back() != '}' || c == std::string::npos || c < 2)
    throw std::runtime_error(
        "Invalid string format representing a pair. Correct format is {identifier:value}. String provided: " + val);
  return {val.substr(1, c - 1), val.substr(c + 1, val.size() - c - 2)};
}

template <>
std::map<std::string, VectorXd> converter<std::map<std::string, VectorXd>>::convert(const std::string& val) {
  if (!is_list_string(val))
    throw std::runtime_error(
        "Invalid string representation of a Map. Correct format is [({id:value})?(,{id:value})*]. String provided: " + val);
  std::map<std::string, VectorXd> m;
  for (const auto& w : parseList(val)) {
    auto p = split_pair(w);
    m.insert({p.first, converter<VectorXd>::convert(p.second)});
  }
  return m;
}

template <>
std::vector<std::map<std::string, std::string>> converter<std::vector<std::map<std::string, std::string>>>::convert(
    const std::string& val) {
  if (!is_list_string(val))
    throw std::runtime_error(
        "Invalid string format representing a list-like structure. Correct format is [(value)?(,value)*]. String "
        "provided: " +
        val);
  std::vector<std::map<std::string, std::string>> out;
  for (const auto& item : parseList(val)) {
    std::map<std::string, std::string> m;
    for (const auto& w : parseList(item)) {
      auto p = split_pair(w);
      m.insert(p);
    }
    out.push_back(m);
  }
  return out;
}

// ---------------------------------------------------------------------------------------------------
// ParserYaml (restates src/utils/parser_yaml.cpp)
// ---------------------------------------------------------------------------------------------------
using yaml_lite::Node;

static std::string parseAtomicNode(const Node& node);

static std::string mapToString(const std::map<std::string, std::string>& map) {  // :77-91
  std::string acc;
  for (const auto& p : map) acc += "{" + p.first + ":" + p.second + "},";
  if (acc.size() > 1)
    acc = acc.substr(0, acc.size() - 1);
  else
    acc = "";
  return "[" + acc + "]";
}
static std::map<std::string, std::string> fetchAsMap(const Node& node) {  // :20-52
  std::map<std::string, std::string> m;
  for (const auto& kv : node.map) {
    std::string key = kv.first;
    switch (kv.second.type) {
      case Node::Scalar:
        m.insert({key, kv.second.scalar});
        break;
      case Node::Sequence:
        m.insert({key, parseAtomicNode(kv.second)});
        break;
      case Node::Map:
        if (!key.empty() && key[0] == '$') key = key.substr(1);
        m.insert({key, mapToString(fetchAsMap(kv.second))});
        break;
      default:
        break;
    }
  }
  return m;
}
static std::string parseAtomicNode(const Node& node) {  // :93-117
  switch (node.type) {
    case Node::Scalar:
      return node.scalar;
    case Node::Sequence: {
      std::string aux;
      bool first = true;
      for (const auto& it : node.seq) {
        aux += (first ? "" : ",") + parseAtomicNode(it);
        first = false;
      }
      return "[" + aux + "]";
    }
    case Node::Map:
      return mapToString(fetchAsMap(node));
    default:
      return "";
  }
}
static bool isAtomic(const std::string& key, const Node& node) {  // :119-156
  switch (node.type) {
    case Node::Scalar:
      return true;
    case Node::Sequence: {
      bool atomic = true;
      for (const auto& it : node.seq) {
        if (it.type == Node::Map) {
          for (const auto& kv : it.map) atomic = atomic && isAtomic(kv.first, it);
        } else {
          atomic = atomic && isAtomic("", it);
        }
      }
      return atomic;
    }
    case Node::Map:
      return !key.empty() && key[0] == '$';
    default:
      throw std::runtime_error("Cannot determine atomicity of node type " + std::to_string((int)node.type));
  }
}

ParserYaml::ParserYaml(const std::string& file, const std::string& path_root, bool freely_parse) : file_(file) {
  if (path_root != "") {
    path_root_ = path_root;
    size_t e = path_root_.size();
    while (e > 0 && path_root_[e - 1] == ' ') --e;
    if (e == 0 || path_root_[e - 1] != '/') path_root_ += "/";
  }
  if (!freely_parse)
    parse();
  else
    parseFreely();
}

std::string ParserYaml::generatePath(const std::string& file) const {
  if (!file.empty() && file[0] == '/') return file;
  return path_root_ + file;
}

void ParserYaml::parse() {  // :191-221
  parseFirstLevel(file_);
  if (robot_.defined()) walkTreeRecursive(robot_, "robot");
  if (is_trajectory_) {
    if (problem_params_.type == Node::Map) walkTreeRecursive(problem_params_, "problem_params");
    for (const auto& stage : stages_) {
      insertRegister("stages/" + stage.name + "/name", stage.name);
      insertRegister("stages/" + stage.name + "/duration", stage.duration);
      insertRegister("stages/" + stage.name + "/transition", stage.transition);
      for (const auto& cost : stage.costs.seq)
        walkTreeRecursive(cost, "stages/" + stage.name + "/costs/" + cost["name"].scalar);
      for (const auto& contact : stage.contacts.seq)
        walkTreeRecursive(contact, "stages/" + stage.name + "/contacts/" + contact["name"].scalar);
    }
  }
}

void ParserYaml::parseFirstLevel(const std::string& file) {  // :223-244
  const Node n = yaml_lite::load_file(generatePath(file));
  const Node& n_trajectory = n["trajectory"];
  if (n_trajectory.type != Node::Map) {
    const Node& n_mpc = n["mpc_controller"];
    if (n_mpc.type != Node::Map) {
      throw std::runtime_error(
          "Could not find neither a trajectory or an mpc_controller node. Please make sure that your YAML file " +
          generatePath(file) + " starts with 'trajectory:' or 'mpc_controller:'");
    }
    is_trajectory_ = false;
    parseMpcController(n_mpc, file);
  } else {
    is_trajectory_ = true;
    parseTrajectory(n_trajectory, file);
  }
}

static std::string list_to_string(const std::vector<std::string>& v) {
  std::string r;
  for (const auto& s : v) r += "," + s;
  if (!r.empty()) r = r.substr(1);
  return "[" + r + "]";
}

void ParserYaml::parseTrajectory(const Node& node, const std::string& file) {  // :246-315
  if (node["robot"].type != Node::Map)
    throw std::runtime_error("Could not find robot node. Please make sure that the 'trajectory' node in YAML file " +
                             generatePath(file) + " has a 'robot' entry");
  robot_ = node["robot"];
  if (node["problem_params"].type == Node::Map) problem_params_ = node["problem_params"];
  if (node["initial_state"].type == Node::Sequence) insertRegister("initial_state", parseAtomicNode(node["initial_state"]));

  std::vector<std::string> stage_strings;
  const Node& stages = node["stages"];
  bool ok = stages.type == Node::Sequence;
  if (ok) {
    for (const auto& stage : stages.seq) {
      if (stage.type != Node::Map || stage["name"].type != Node::Scalar || stage["duration"].type != Node::Scalar ||
          stage["costs"].type != Node::Sequence) {
        ok = false;
        break;
      }
      // 'transition' is "true" whenever the key exists, whatever its value (:274-278)
      const std::string transition = stage["transition"].defined() ? "true" : "false";
      StageInit p{stage["name"].scalar, stage["duration"].scalar, transition, stage["costs"], stage["contacts"]};
      stages_.push_back(p);
      std::vector<std::string> cost_names, contact_names;
      for (const auto& c : p.costs.seq) cost_names.push_back(c["name"].scalar);
      for (const auto& c : p.contacts.seq) contact_names.push_back(c["name"].scalar);
      std::map<std::string, std::string> m = {{"name", p.name},
                                              {"duration", p.duration},
                                              {"transition", transition},
                                              {"costs", list_to_string(cost_names)}};
      if (p.contacts.defined()) m.insert({"contacts", list_to_string(contact_names)});
      stage_strings.push_back(mapToString(m));
    }
  }
  if (!ok)
    throw std::runtime_error("Error parsing stages @" + generatePath(file) +
                             ". Make sure every stage has a name, duration and, at least, one cost.");
  insertRegister("stages", list_to_string(stage_strings));
}

void ParserYaml::parseMpcController(const Node& node, const std::string& file) {  // :317-334
  if (node["robot"].type != Node::Map)
    throw std::runtime_error("Could not find robot node. Please make sure that the 'trajectory' node in YAML file " +
                             generatePath(file) + " has a 'robot' entry");
  for (const auto& kv : node.map) {
    if (kv.first == "robot")
      robot_ = kv.second;
    else
      insertRegister("mpc_controller/" + kv.first, parseAtomicNode(kv.second));
  }
}

void ParserYaml::parseFreely() { walkTreeRecursive(yaml_lite::load_file(file_), ""); }

void ParserYaml::walkTreeFile(const std::string& file, const std::string& node_name) {
  walkTreeRecursive(yaml_lite::load_file(generatePath(file)), node_name);
}

void ParserYaml::walkTreeRecursive(const Node& node, const std::string& node_name) {  // :363-442
  switch (node.type) {
    case Node::Scalar:
      if (!node.scalar.empty() && node.scalar[0] == '@')
        walkTreeFile(node.scalar.substr(1), node_name);
      else
        insertRegister(node_name, node.scalar);
      break;
    case Node::Sequence:
      if (isAtomic("", node)) {
        insertRegister(node_name, parseAtomicNode(node));
      } else {
        for (const auto& it : node.seq) walkTreeRecursive(it, node_name);
      }
      break;
    case Node::Map:
      for (const auto& kv : node.map) {
        if (isAtomic(kv.first, node)) {
          insertRegister(node_name + "/" + kv.first.substr(1), parseAtomicNode(kv.second));
        } else if (kv.first != "follow") {
          walkTreeRecursive(kv.second, node_name + "/" + kv.first);
        } else {
          // 'follow' splices another file in at the same depth (:415-431)
          walkTreeFile(getYamlPath(kv.second.scalar), node_name);
        }
      }
      break;
    default:
      break;
  }
}

void ParserYaml::insertRegister(std::string key, const std::string& value) {  // :470-480
  if (key.substr(0, 1) == "/") key = key.substr(1);
  auto inserted = params_.insert({key, value});
  if (!inserted.second)
    std::cout << "Skipping key '" << key << "' with value '" << value << "'. There already exists the register: ("
              << inserted.first->first << "," << inserted.first->second << ")" << std::endl;
}

}  // namespace eagle_mpc
