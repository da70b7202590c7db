#include "robot_model.hpp"

#include <cmath>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>
#include <stdexcept>

namespace eagle_mpc {

namespace {

// ---- minimal XML reader (elements + attributes; text, comments, declarations skipped) ----------------
struct Xml {
  std::string name;
  std::map<std::string, std::string> attr;
  std::vector<Xml> children;
  const Xml* child(const std::string& n) const {
    for (const auto& c : children)
      if (c.name == n) return &c;
    return nullptr;
  }
};

struct XmlParser {
  const std::string& s;
  size_t i = 0;
  explicit XmlParser(const std::string& text) : s(text) {}
  [[noreturn]] void fail(const std::string& m) { throw std::runtime_error("urdf: " + m + " at offset " + std::to_string(i)); }
  void skip_ws() {
    while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
  }
  bool starts(const char* t) const { return s.compare(i, std::strlen(t), t) == 0; }
  void skip_misc() {
    while (true) {
      skip_ws();
      if (starts("<!--")) {
        size_t e = s.find("-->", i);
        if (e == std::string::npos) fail("unterminated comment");
        i = e + 3;
      } else if (starts("<?")) {
        size_t e = s.find("?>", i);
        if (e == std::string::npos) fail("unterminated declaration");
        i = e + 2;
      } else if (starts("<!")) {
        size_t e = s.find(">", i);
        if (e == std::string::npos) fail("unterminated doctype");
        i = e + 1;
      } else if (i < s.size() && s[i] != '<') {
        while (i < s.size() && s[i] != '<') ++i;  // text content
      } else {
        return;
      }
    }
  }
  std::string ident() {
    size_t b = i;
    while (i < s.size() && (std::isalnum((unsigned char)s[i]) || s[i] == '_' || s[i] == ':' || s[i] == '-' || s[i] == '.')) ++i;
    if (b == i) fail("identifier expected");
    return s.substr(b, i - b);
  }
  Xml element() {
    skip_misc();
    if (i >= s.size() || s[i] != '<') fail("'<' expected");
    ++i;
    Xml e;
    e.name = ident();
    while (true) {
      skip_ws();
      if (starts("/>")) {
        i += 2;
        return e;
      }
      if (i < s.size() && s[i] == '>') {
        ++i;
        break;
      }
      std::string key = ident();
      skip_ws();
      if (i >= s.size() || s[i] != '=') fail("'=' expected");
      ++i;
      skip_ws();
      if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) fail("quote expected");
      const char q = s[i++];
      size_t e2 = s.find(q, i);
      if (e2 == std::string::npos) fail("unterminated attribute");
      e.attr[key] = s.substr(i, e2 - i);
      i = e2 + 1;
    }
    while (true) {
      skip_misc();
      if (starts("</")) {
        i += 2;
        std::string n = ident();
        if (n != e.name) fail("mismatched closing tag " + n);
        skip_ws();
        if (i >= s.size() || s[i] != '>') fail("'>' expected");
        ++i;
        return e;
      }
      if (i >= s.size()) fail("unexpected end of file");
      e.children.push_back(element());
    }
  }
};

std::vector<double> numbers(const std::string& s, size_t n, const std::string& what) {
  std::istringstream ss(s);
  std::vector<double> v;
  double d;
  while (ss >> d) v.push_back(d);
  if (v.size() != n) throw std::runtime_error("urdf: attribute " + what + " needs " + std::to_string(n) + " numbers");
  return v;
}

void rpy_to_R(const double* rpy, double* R) {
  const double cr = std::cos(rpy[0]), sr = std::sin(rpy[0]);
  const double cp = std::cos(rpy[1]), sp = std::sin(rpy[1]);
  const double cy = std::cos(rpy[2]), sy = std::sin(rpy[2]);
  // R = Rz(yaw) Ry(pitch) Rx(roll)
  R[0] = cy * cp;
  R[1] = cy * sp * sr - sy * cr;
  R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp;
  R[4] = sy * sp * sr + cy * cr;
  R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;
  R[7] = cp * sr;
  R[8] = cp * cr;
}
void mul33(const double* a, const double* b, double* r) {
  double t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  std::memcpy(r, t, sizeof(t));
}
void mulT33(const double* a, const double* b, double* r) {  // a b^T
  double t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = a[3 * i] * b[3 * j] + a[3 * i + 1] * b[3 * j + 1] + a[3 * i + 2] * b[3 * j + 2];
  std::memcpy(r, t, sizeof(t));
}
void mul31(const double* a, const double* v, double* r) {
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2];
  std::memcpy(r, t, sizeof(t));
}

struct Pose {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double p[3] = {0, 0, 0};
};
Pose compose(const Pose& a, const Pose& b) {
  Pose r;
  mul33(a.R, b.R, r.R);
  double t[3];
  mul31(a.R, b.p, t);
  for (int i = 0; i < 3; ++i) r.p[i] = a.p[i] + t[i];
  return r;
}
Pose origin_of(const Xml* parent) {
  Pose o;
  const Xml* e = parent ? parent->child("origin") : nullptr;
  if (!e) return o;
  if (e->attr.count("xyz")) {
    auto v = numbers(e->attr.at("xyz"), 3, "origin xyz");
    for (int i = 0; i < 3; ++i) o.p[i] = v[i];
  }
  if (e->attr.count("rpy")) {
    auto v = numbers(e->attr.at("rpy"), 3, "origin rpy");
    rpy_to_R(v.data(), o.R);
  }
  return o;
}

// add a rigid body (m2, c2, I2 about its COM) to a composite (m, c, I about its COM); all in one frame
void add_inertia(double& m, double* c, double* I, double m2, const double* c2, const double* I2) {
  if (m2 <= 0) return;
  const double M = m + m2;
  double cn[3];
  for (int i = 0; i < 3; ++i) cn[i] = (m * c[i] + m2 * c2[i]) / M;
  auto shift = [&](double mass, const double* from, double* acc) {
    double d[3] = {from[0] - cn[0], from[1] - cn[1], from[2] - cn[2]};
    const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) acc[3 * i + j] += mass * ((i == j ? dd : 0.0) - d[i] * d[j]);
  };
  double In[9];
  for (int i = 0; i < 9; ++i) In[i] = I[i] + I2[i];
  shift(m, c, In);
  shift(m2, c2, In);
  m = M;
  for (int i = 0; i < 3; ++i) c[i] = cn[i];
  std::memcpy(I, In, sizeof(In));
}

}  // namespace

RobotModel RobotModel::fromUrdf(const std::string& path) {
  std::ifstream f(path);
  if (!f.good()) throw std::runtime_error("urdf: could not open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return fromUrdfString(ss.str());
}

RobotModel RobotModel::fromUrdfString(const std::string& xml) {
  XmlParser xp(xml);
  Xml robot = xp.element();
  if (robot.name != "robot") throw std::runtime_error("urdf: root element must be <robot>");

  std::map<std::string, const Xml*> links;
  std::vector<const Xml*> joints;
  std::map<std::string, bool> is_child;
  for (const auto& e : robot.children) {
    if (e.name == "link") {
      if (!e.attr.count("name")) throw std::runtime_error("urdf: <link> without name");
      links[e.attr.at("name")] = &e;
    } else if (e.name == "joint") {
      joints.push_back(&e);
    }
  }
  for (const Xml* j : joints) {
    const Xml* c = j->child("child");
    const Xml* p = j->child("parent");
    if (!c || !p || !c->attr.count("link") || !p->attr.count("link")) throw std::runtime_error("urdf: joint without parent/child");
    if (!links.count(c->attr.at("link")) || !links.count(p->attr.at("link")))
      throw std::runtime_error("urdf: joint " + j->attr.at("name") + " references an unknown link");
    is_child[c->attr.at("link")] = true;
  }
  std::string root;
  for (const auto& e : robot.children)
    if (e.name == "link" && !is_child[e.attr.at("name")]) {
      if (!root.empty()) throw std::runtime_error("urdf: more than one root link");
      root = e.attr.at("name");
    }
  if (root.empty()) throw std::runtime_error("urdf: no root link");

  RobotModel M;
  EmpcModelDesc& d = M.desc_;
  std::memset(&d, 0, sizeof(d));
  d.nbodies = 1;
  d.parent[0] = -1;
  d.gravity[0] = 0;
  d.gravity[1] = 0;
  d.gravity[2] = -9.81;
  for (int b = 0; b < EMPC_MAX_BODIES; ++b) {
    d.jplace_R[b][0] = d.jplace_R[b][4] = d.jplace_R[b][8] = 1;
  }
  M.effort_limit_.assign(6, 0.0);

  std::function<void(const std::string&, int, const Pose&)> visit = [&](const std::string& link_name, int body,
                                                                       const Pose& link_in_body) {
    const Xml* link = links.at(link_name);
    // frame
    RobotFrame fr;
    fr.name = link_name;
    fr.body = body;
    std::memcpy(fr.R, link_in_body.R, sizeof(fr.R));
    std::memcpy(fr.p, link_in_body.p, sizeof(fr.p));
    M.frames_.push_back(fr);
    // inertial
    if (const Xml* in = link->child("inertial")) {
      const Xml* me = in->child("mass");
      const double mass = me && me->attr.count("value") ? std::stod(me->attr.at("value")) : 0.0;
      double I[9] = {0};
      if (const Xml* ie = in->child("inertia")) {
        auto g = [&](const char* k) { return ie->attr.count(k) ? std::stod(ie->attr.at(k)) : 0.0; };
        I[0] = g("ixx");
        I[1] = I[3] = g("ixy");
        I[2] = I[6] = g("ixz");
        I[4] = g("iyy");
        I[5] = I[7] = g("iyz");
        I[8] = g("izz");
      }
      const Pose io = compose(link_in_body, origin_of(in));
      double RI[9], Ib[9];
      mul33(io.R, I, RI);
      mulT33(RI, io.R, Ib);
      add_inertia(d.mass[body], d.com[body], d.inertia[body], mass, io.p, Ib);
    }
    // children, in file order
    for (const Xml* j : joints) {
      if (j->child("parent")->attr.at("link") != link_name) continue;
      const std::string type = j->attr.count("type") ? j->attr.at("type") : "";
      const Pose jp = compose(link_in_body, origin_of(j));
      const std::string child = j->child("child")->attr.at("link");
      if (type == "fixed") {
        visit(child, body, jp);
      } else if (type == "revolute" || type == "continuous") {
        if (d.nbodies >= EMPC_MAX_BODIES) throw std::runtime_error("urdf: too many moving joints");
        const int b = d.nbodies++;
        d.parent[b] = body;
        std::memcpy(d.jplace_R[b], jp.R, sizeof(jp.R));
        std::memcpy(d.jplace_p[b], jp.p, sizeof(jp.p));
        double ax[3] = {1, 0, 0};
        if (const Xml* a = j->child("axis"))
          if (a->attr.count("xyz")) {
            auto v = numbers(a->attr.at("xyz"), 3, "axis xyz");
            const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            if (n <= 0) throw std::runtime_error("urdf: zero joint axis");
            for (int i = 0; i < 3; ++i) ax[i] = v[i] / n;
          }
        std::memcpy(d.axis[b], ax, sizeof(ax));
        double effort = 0;
        if (const Xml* l = j->child("limit"))
          if (l->attr.count("effort")) effort = std::stod(l->attr.at("effort"));
        d.effort_limit[b] = effort;
        M.effort_limit_.push_back(effort);
        M.joint_names_.push_back(j->attr.count("name") ? j->attr.at("name") : "");
        visit(child, b, Pose());
      } else {
        throw std::runtime_error("urdf: unsupported joint type '" + type + "' (fixed, revolute, continuous only)");
      }
    }
  };
  visit(root, 0, Pose());
  d.nq = 7 + d.nbodies - 1;
  d.nv = 6 + d.nbodies - 1;
  d.nframes = 0;
  return M;
}

std::size_t RobotModel::getFrameId(const std::string& name) const {
  for (std::size_t i = 0; i < frames_.size(); ++i)
    if (frames_[i].name == name) return i;
  return frames_.size();
}

double RobotModel::totalMass() const {
  double m = 0;
  for (int b = 0; b < desc_.nbodies; ++b) m += desc_.mass[b];
  return m;
}

EmpcModelDesc RobotModel::descWithFrames(const std::vector<int>& frame_ids) const {
  EmpcModelDesc d = desc_;
  if (frame_ids.size() > EMPC_MAX_FRAMES) throw std::runtime_error("too many operational frames referenced by the problem");
  d.nframes = (int)frame_ids.size();
  for (int i = 0; i < d.nframes; ++i) {
    const RobotFrame& f = frames_.at(frame_ids[i]);
    d.frame_body[i] = f.body;
    std::memcpy(d.frame_R[i], f.R, sizeof(f.R));
    std::memcpy(d.frame_p[i], f.p, sizeof(f.p));
    std::strncpy(d.frame_name[i], f.name.c_str(), EMPC_NAME_LEN - 1);
  }
  return d;
}

}  // namespace eagle_mpc
