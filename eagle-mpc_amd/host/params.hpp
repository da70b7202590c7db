// params.hpp: ParserYaml, ParamsServer and the string converters.
//
// Mirrors the reference's flat-key parameter machinery so that every factory reads exactly the keys the
// reference reads:
//   ParserYaml   include/eagle_mpc/utils/parser_yaml.hpp:26-31, src/utils/parser_yaml.cpp:176-482
//   ParamsServer include/eagle_mpc/utils/params_server.hpp:26-69 (getParam<T> throws MissingValueException)
//   converter<T> include/eagle_mpc/utils/converter.hpp:27-263 (stoi/stod scalars, literal true/false bools,
//                vectors restricted to -?[0-9]*(\.[0-9]+)? elements: src/utils/converter_utils.cpp:39-40)
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "yaml_lite.hpp"

namespace eagle_mpc {

typedef std::vector<double> VectorXd;

class MissingValueException : public std::runtime_error {
 public:
  explicit MissingValueException(const std::string& msg) : std::runtime_error(msg) {}
};

// Data directories (the reference bakes EAGLE_MPC_YAML_DIR / EAGLE_MPC_ROBOT_DATA_DIR in at configure time,
// config/path.hpp.in:4-11; here they are runtime settings, defaulting to <package>/data/{yaml,robots}).
void set_yaml_dir(const std::string& dir);
void set_robot_data_dir(const std::string& dir);
const std::string& yaml_dir();
const std::string& robot_data_dir();
std::string getYamlPath(const std::string& yaml_path);  // src/utils/parser_yaml.cpp:158-163
std::string getUrdfPath(const std::string& urdf_path);  // src/utils/parser_yaml.cpp:165-170

template <typename T>
struct converter {
  static T convert(const std::string& val);
};
template <>
int converter<int>::convert(const std::string& val);
template <>
double converter<double>::convert(const std::string& val);
template <>
bool converter<bool>::convert(const std::string& val);
template <>
std::string converter<std::string>::convert(const std::string& val);
template <>
VectorXd converter<VectorXd>::convert(const std::string& val);
template <>
std::vector<std::string> converter<std::vector<std::string>>::convert(const std::string& val);
template <>
std::map<std::string, VectorXd> converter<std::map<std::string, VectorXd>>::convert(const std::string& val);
template <>
std::vector<std::map<std::string, std::string>> converter<std::vector<std::map<std::string, std::string>>>::convert(
    const std::string& val);

// split the top-level items of "[a,b,[c,d],{e:f}]" (reference utils::parseList)
std::vector<std::string> parseList(const std::string& val);

class ParamsServer {
 public:
  ParamsServer() {}
  explicit ParamsServer(const std::map<std::string, std::string>& params) : params_(params) {}
  void addParam(const std::string& key, const std::string& value) { params_.insert({key, value}); }
  bool hasParam(const std::string& key) const { return params_.find(key) != params_.end(); }
  template <typename T>
  T getParam(const std::string& key) const {
    auto it = params_.find(key);
    if (it == params_.end())
      throw MissingValueException("The following key: '" + key + "' has not been found in the parameters server.");
    return converter<T>::convert(it->second);
  }
  const std::map<std::string, std::string>& get_params() const { return params_; }

 private:
  std::map<std::string, std::string> params_;
};

class ParserYaml {
 public:
  explicit ParserYaml(const std::string& file, const std::string& path_root = "", bool freely_parse = false);
  const std::map<std::string, std::string>& get_params() const { return params_; }

 private:
  struct StageInit {
    std::string name, duration, transition;
    yaml_lite::Node costs, contacts;
  };
  void parse();
  void parseFreely();
  void parseFirstLevel(const std::string& file);
  void parseTrajectory(const yaml_lite::Node& node, const std::string& file);
  void parseMpcController(const yaml_lite::Node& node, const std::string& file);
  void walkTreeRecursive(const yaml_lite::Node& node, const std::string& node_name);
  void walkTreeFile(const std::string& file, const std::string& node_name);
  void insertRegister(std::string key, const std::string& value);
  std::string generatePath(const std::string& file) const;

  std::string file_, path_root_;
  bool is_trajectory_ = false;
  yaml_lite::Node robot_, problem_params_;
  std::vector<StageInit> stages_;
  std::map<std::string, std::string> params_;
};

}  // namespace eagle_mpc
