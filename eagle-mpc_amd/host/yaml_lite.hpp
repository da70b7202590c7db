// yaml_lite: the YAML subset the eagle-mpc problem files use: block mappings,
// block sequences of mappings, flow sequences of scalars (possibly starting on the next line or spanning
// lines), plain / quoted scalars and '#' comments.  No anchors, tags, multi-documents or block scalars.
// Stands in for yaml-cpp (reference: src/utils/parser_yaml.cpp uses YAML::Node / YAML::LoadFile), which is
// not available on the target image.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace eagle_mpc {
namespace yaml_lite {

struct Node {
  enum Type { Undefined, Null, Scalar, Sequence, Map };
  Type type = Undefined;
  std::string scalar;
  std::vector<Node> seq;
  std::vector<std::pair<std::string, Node>> map;  // insertion order, like yaml-cpp iteration

  bool defined() const { return type != Undefined; }
  // map lookup; returns an Undefined node when the key is missing (yaml-cpp semantics of node["key"])
  const Node& operator[](const std::string& key) const;
};

Node load_string(const std::string& text);
Node load_file(const std::string& path);  // throws std::runtime_error("Couldn't load file: ...")

}  // namespace yaml_lite
}  // namespace eagle_mpc
