// graph.hip -- simple_graph.sgh reader / writer (host code only).
//
// Replaces (cfear_radarodometry/):
//   SaveSimpleGraph / LoadSimpleGraph                 src/cfear_radarodometry/types.cpp:103-130
//   RadarScan / Pose3d / Constraint3d ::serialize      include/cfear_radarodometry/types.h:46-192
//   MapPointNormal::save / load, cell::serialize       include/cfear_radarodometry/pointnormal.h:88-101, 206-228
//   the free serialize() functions                     include/cfear_radarodometry/serialization.h
//
// The file is a Boost.Serialization binary_oarchive (Boost 1.71, archive version 17, the version of the reference's
// Docker image, tbv_slam/docker/Dockerfile) of
//     typedef std::vector<std::pair<RadarScan, std::vector<Constraint3d>>> simple_graph;
// Boost itself is not in this image, so the archive's byte layout is restated here from the library's published
// sources (basic_binary_oarchive.hpp, basic_oarchive.cpp, oserializer.hpp, collections_save_imp.hpp, shared_ptr.hpp):
//   * header: the string "serialization::archive" (u64 length + bytes), the library version as u16, then the native sizes
//     of int / long / float / double as bytes (4 8 4 8) and int32 1 (basic_binary_oprimitive::init);
//   * strings: u64 length + bytes; bool: 1 byte; enums: int32; collection sizes: u64; item_version: u32;
//   * every CLASS type (anything that is not a primitive) writes, the first time an object of it is saved,
//     tracking_type (1 byte) + version_type (u32); types serialized through a pointer are "tracked": the first
//     occurrence writes class_id (i16) [+ tracking + version], later ones class_id_reference (i16), followed by an
//     object_id (u32) -- a NEW id is followed by the object's data, a known id is a back-reference (this is how
//     cloud_normal_->input_ and cloud_nopeaks_, one shared PointCloud, are stored once); a null pointer is class_id -1;
//   * class ids count EVERY class type in order of first appearance, so they depend on the data (an empty first cloud
//     delays PointXYZI): reader and writer below share one traversal and assign ids on the fly;
//   * C arrays (PointXYZI::data) are a u64 count + raw elements; boost::serialization::make_array (Eigen matrices) is
//     raw elements without a count.
// NO reference-produced .sgh exists in the repository (its test refers to a file that is not committed), so this layout
// is UNPINNED against a real file: what is tested is the round trip and the byte layout of hand-checked small cases.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/cfear_hip.h"

namespace {

struct Cloud {
  bool present = false;
  std::vector<float> xyzi;           // [n][4]
  uint64_t stamp = 0;
  uint32_t seq = 0;
  std::string frame_id;
  const float* src = nullptr;        // saving: the caller's buffer -- the same buffer in two slots is ONE shared cloud
};
struct ConstraintRec {
  uint64_t id_begin = 0, id_end = 0;
  cfear_pose3d t_be{};
  double information[36] = {0};
  int32_t type = 0;
  std::vector<std::string> qkeys;
  std::vector<double> qvals;
  std::string info;
  std::vector<const char*> qkey_ptrs;
};
struct NodeRec {
  cfear_pose3d T{}, Tgt{};
  bool has_Tgt = false;
  uint32_t idx = 0;
  uint64_t stamp = 0;
  double motion[16] = {0};
  Cloud peaks, nopeaks, input;       // input: cloud_normal_->input_ when it is NOT the nopeaks object
  bool has_normal = false, input_is_nopeaks = true;
  std::vector<cfear_cell> cells;
  float radius = 0.f;
  bool weight_intensity = false;
  std::vector<ConstraintRec> constraints;
  std::vector<cfear_graph_constraint> constraint_views;
};

// ---- class / object bookkeeping shared by both directions -----------------------------------------------------
enum ClassKey { K_GRAPH, K_PAIR, K_SCAN, K_POSE, K_VEC3, K_QUAT, K_AFF3, K_SP_CLOUDI, K_CLOUDI, K_PXYZI, K_SP_MAP, K_MAP,
                K_VCELL, K_CELL, K_VEC2, K_MAT2, K_SP_CLOUDXY, K_CLOUDXY, K_PXY, K_VCONS, K_CONS, K_MAT6, K_QMAP, K_QPAIR, K_COUNT };
struct ClassInfo { uint8_t tracking; uint32_t version; };
const ClassInfo kInfo[K_COUNT] = {
    {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0},
    {0, 1},   // boost::shared_ptr<T>: version 1, track_never
    {1, 0},   // pcl::PointCloud<PointXYZI>, saved through a pointer -> tracked
    {0, 0},
    {0, 1}, {1, 0},   // shared_ptr<MapPointNormal>, MapPointNormal
    {0, 0}, {0, 0}, {0, 0}, {0, 0},
    {0, 1}, {1, 0}, {0, 0},   // shared_ptr<PointCloud<PointXY>>, PointCloud<PointXY>, PointXY
    {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};

struct Archive {
  FILE* f = nullptr;
  bool saving = false, ok = true;
  int class_id[K_COUNT];
  bool initialized[K_COUNT];
  int n_classes = 0;
  uint32_t n_objects = 0;
  // loading: what is left of the file bounds every size the file dictates; clouds by object id, so that a pointer that
  // refers back to a cloud stored earlier (boost tracks shared_ptr targets) comes back as that cloud
  uint64_t file_bytes = 0;
  std::vector<std::pair<uint32_t, const struct Cloud*>> clouds;   // (saving: the clouds written so far)
  bool fits(uint64_t count, uint64_t min_bytes_each) {
    const long at = ftell(f);
    if (at < 0 || (uint64_t)at > file_bytes || count > (file_bytes - (uint64_t)at) / (min_bytes_each ? min_bytes_each : 1)) { ok = false; return false; }
    return true;
  }
  Archive() { for (int i = 0; i < K_COUNT; i++) { class_id[i] = -1; initialized[i] = false; } }
  void raw(void* p, size_t n) {
    if (!ok || n == 0) return;
    ok = saving ? fwrite(p, 1, n, f) == n : fread(p, 1, n, f) == n;
  }
  template <typename T> void prim(T& v) { raw(&v, sizeof(T)); }
  void str(std::string& s) {
    uint64_t n = s.size();
    prim(n);
    if (!saving) { if (n > (1ull << 30)) { ok = false; return; } s.resize((size_t)n); }
    if (n) raw(&s[0], (size_t)n);
  }
  int reg(ClassKey k) { if (class_id[k] < 0) class_id[k] = n_classes++; return class_id[k]; }
  // object saved by value: class info (tracking, version) the first time the class appears
  void object_preamble(ClassKey k) {
    reg(k);
    if (!initialized[k]) {
      uint8_t t = kInfo[k].tracking; uint32_t v = kInfo[k].version;
      prim(t); prim(v);
      if (!saving && (t != kInfo[k].tracking)) ok = false;
      initialized[k] = true;
    }
  }
  // object saved through a pointer.  Returns 0 = null, 1 = new object (data follows), 2 = back-reference (object id in *oid)
  int pointer_preamble(ClassKey k, bool is_null, uint32_t known_id, bool have_known, uint32_t* oid) {
    if (saving) {
      if (is_null) { int16_t c = -1; prim(c); return 0; }
      reg(k);
      int16_t c = (int16_t)class_id[k];
      prim(c);
      if (!initialized[k]) { uint8_t t = kInfo[k].tracking; uint32_t v = kInfo[k].version; prim(t); prim(v); initialized[k] = true; }
      if (have_known) { uint32_t o = known_id; prim(o); *oid = o; return 2; }
      uint32_t o = n_objects++;
      prim(o); *oid = o;
      return 1;
    }
    int16_t c = 0;
    prim(c);
    if (!ok) return 0;
    if (c == -1) return 0;
    if (class_id[k] < 0) {
      if (c != n_classes) { ok = false; return 0; }       // a class seen for the first time takes the next id
      reg(k);
    } else if (c != class_id[k]) { ok = false; return 0; }
    if (!initialized[k]) { uint8_t t = 0; uint32_t v = 0; prim(t); prim(v); initialized[k] = true; }
    uint32_t o = 0;
    prim(o);
    *oid = o;
    if (o < n_objects) return 2;
    if (o != n_objects) { ok = false; return 0; }
    n_objects++;
    return 1;
  }
};

void io_vec3(Archive& a, double p[3]) { a.object_preamble(K_VEC3); for (int i = 0; i < 3; i++) a.prim(p[i]); }
void io_vec2(Archive& a, double p[2]) { a.object_preamble(K_VEC2); for (int i = 0; i < 2; i++) a.prim(p[i]); }
void io_pose(Archive& a, cfear_pose3d& p) {
  a.object_preamble(K_POSE);
  io_vec3(a, p.p);
  a.object_preamble(K_QUAT);
  for (int i = 0; i < 4; i++) a.prim(p.q[i]);          // x, y, z, w (serialization.h:44-50)
}
void io_count(Archive& a, uint64_t& n, ClassKey /*of*/) {
  a.prim(n);
  uint32_t item_version = 0;
  a.prim(item_version);
}

// pcl::PointCloud<PointXYZI>: serialization.h:101-111, 122-126
void io_cloud_xyzi_data(Archive& a, Cloud& c) {
  a.prim(c.stamp); a.prim(c.seq); a.str(c.frame_id);
  uint32_t n = (uint32_t)(c.xyzi.size() / 4), height = 1, width = n;
  a.prim(height); a.prim(width);
  if (!a.saving) {
    const uint64_t n64 = (uint64_t)height * width;
    if (n64 > (1u << 26) || !a.fits(n64, 28)) { a.ok = false; return; }    // a point: count (8) + data[4] (16) + intensity (4)
    n = (uint32_t)n64;
    c.xyzi.assign((size_t)n * 4, 0.f);
  }
  for (uint32_t i = 0; i < n && a.ok; i++) {
    a.object_preamble(K_PXYZI);
    uint64_t cnt = 4;                                   // float data[4]: C array = count + raw elements
    a.prim(cnt);
    float d[4] = {c.xyzi[4 * i], c.xyzi[4 * i + 1], c.xyzi[4 * i + 2], 1.0f};
    a.raw(d, 16);
    float inten = c.xyzi[4 * i + 3];
    a.prim(inten);
    if (!a.saving) { c.xyzi[4 * i] = d[0]; c.xyzi[4 * i + 1] = d[1]; c.xyzi[4 * i + 2] = d[2]; c.xyzi[4 * i + 3] = inten; }
  }
}

// shared_ptr<PointCloud<PointXYZI>>; objects[] maps object ids to clouds already stored
int io_cloud_ptr(Archive& a, Cloud& c, const Cloud* same_as, uint32_t same_oid, bool has_same, uint32_t* oid_out) {
  a.object_preamble(K_SP_CLOUDI);
  uint32_t oid = 0;
  if (a.saving && !has_same && c.present && c.src) {        // boost writes a tracked object once: the caller's buffer is the identity
    for (const auto& oc : a.clouds)
      if (oc.second->src == c.src && oc.second->xyzi.size() == c.xyzi.size() && oc.second->stamp == c.stamp &&
          oc.second->seq == c.seq && oc.second->frame_id == c.frame_id) {   // (the same buffer under another header is another object)
        same_as = oc.second; same_oid = oc.first; has_same = true; break; }
  }
  const int st = a.pointer_preamble(K_CLOUDI, !c.present && !(has_same && same_as), same_oid, has_same, &oid);
  if (st == 1) {
    c.present = true;
    io_cloud_xyzi_data(a, c);
    a.clouds.push_back({oid, &c});                         // (the node records do not move while the file is read / written)
  } else if (st == 2 && !a.saving) {                         // the same cloud object as one stored earlier
    for (const auto& oc : a.clouds)
      if (oc.first == oid && oc.second != &c) { c = *oc.second; break; }
  }
  if (oid_out) *oid_out = oid;
  return st;
}

void io_node(Archive& a, NodeRec& n) {
  a.object_preamble(K_SCAN);
  io_pose(a, n.T);
  io_pose(a, n.Tgt);
  uint8_t b = n.has_Tgt ? 1 : 0;
  a.prim(b);
  n.has_Tgt = b != 0;
  a.prim(n.idx);
  a.prim(n.stamp);
  a.object_preamble(K_AFF3);
  for (int i = 0; i < 16; i++) a.prim(n.motion[i]);
  uint32_t oid_peaks = 0, oid_nopeaks = 0;
  io_cloud_ptr(a, n.peaks, nullptr, 0, false, &oid_peaks);
  const int st_np = io_cloud_ptr(a, n.nopeaks, nullptr, 0, false, &oid_nopeaks);
  // cloud_normal_
  a.object_preamble(K_SP_MAP);
  uint32_t oid = 0;
  const int st = a.pointer_preamble(K_MAP, !n.has_normal, 0, false, &oid);
  if (st == 1) {
    n.has_normal = true;
    a.object_preamble(K_VCELL);
    uint64_t nc = n.cells.size();
    io_count(a, nc, K_CELL);
    if (!a.saving) { if (nc > (1u << 24) || !a.fits(nc, 113)) { a.ok = false; return; } n.cells.assign((size_t)nc, cfear_cell{}); }   // a cell: 13 doubles + size_t + 1 byte
    for (uint64_t i = 0; i < nc && a.ok; i++) {
      cfear_cell& c = n.cells[(size_t)i];
      a.object_preamble(K_CELL);
      io_vec2(a, c.mean);                                // u_
      a.object_preamble(K_MAT2);
      double m[4] = {c.cov[0], c.cov[2], c.cov[1], c.cov[3]};   // Eigen column-major data()
      a.raw(m, 32);
      if (!a.saving) { c.cov[0] = m[0]; c.cov[2] = m[1]; c.cov[1] = m[2]; c.cov[3] = m[3]; }
      a.prim(c.scale);
      io_vec2(a, c.normal);                              // snormal_
      a.prim(c.lambda_min); a.prim(c.lambda_max);
      double sum_intensity = c.avg_intensity * (double)c.nsamples;   // sum_intensity_ = avg_intensity_ * Nsamples_ (pointnormal.cpp:19)
      a.prim(sum_intensity);
      a.prim(c.avg_intensity);
      uint64_t ns = (uint64_t)c.nsamples;                // size_t Nsamples_
      a.prim(ns);
      c.nsamples = (int32_t)ns;
      uint8_t valid = 1;
      a.prim(valid);
    }
    // input_
    if (a.saving) {
      if (n.input_is_nopeaks && n.nopeaks.present) io_cloud_ptr(a, n.input, &n.nopeaks, oid_nopeaks, true, nullptr);
      else io_cloud_ptr(a, n.input, nullptr, 0, false, nullptr);
    } else {
      uint32_t o = 0;
      const int s2 = io_cloud_ptr(a, n.input, nullptr, 0, false, &o);
      n.input_is_nopeaks = (s2 == 2 && st_np != 0 && o == oid_nopeaks);
      if (s2 == 2 && !n.input_is_nopeaks && !(st_np == 0)) n.input_is_nopeaks = (o == oid_nopeaks);
    }
    // downsampled_: PointCloud<PointXY> of the float cell means (ComputeSearchTreeFromCells, pointnormal.cpp:151-162)
    a.object_preamble(K_SP_CLOUDXY);
    uint32_t od = 0;
    const int sd = a.pointer_preamble(K_CLOUDXY, false, 0, false, &od);
    if (sd == 1) {
      uint64_t stamp = 0; uint32_t seq = 0; std::string frame;
      a.prim(stamp); a.prim(seq); a.str(frame);
      uint32_t height = 1, width = (uint32_t)n.cells.size();
      a.prim(height); a.prim(width);
      const uint32_t np = height * width;
      for (uint32_t i = 0; i < np && a.ok; i++) {
        a.object_preamble(K_PXY);
        float x = i < n.cells.size() ? (float)n.cells[i].mean[0] : 0.f, y = i < n.cells.size() ? (float)n.cells[i].mean[1] : 0.f;
        a.prim(x); a.prim(y);
      }
    }
    a.prim(n.radius);
    uint8_t wi = n.weight_intensity ? 1 : 0;
    a.prim(wi);
    n.weight_intensity = wi != 0;
  }
}

void io_constraint(Archive& a, ConstraintRec& c) {
  a.object_preamble(K_CONS);
  a.prim(c.id_begin); a.prim(c.id_end);
  io_pose(a, c.t_be);
  a.object_preamble(K_MAT6);
  double m[36];
  for (int r = 0; r < 6; r++) for (int q = 0; q < 6; q++) m[q * 6 + r] = c.information[r * 6 + q];   // column-major data()
  a.raw(m, sizeof(m));
  if (!a.saving) for (int r = 0; r < 6; r++) for (int q = 0; q < 6; q++) c.information[r * 6 + q] = m[q * 6 + r];
  a.prim(c.type);
  a.object_preamble(K_QMAP);
  uint64_t nq = c.qkeys.size();
  io_count(a, nq, K_QPAIR);
  if (!a.saving) { if (nq > (1u << 20) || !a.fits(nq, 16)) { a.ok = false; return; } c.qkeys.assign((size_t)nq, std::string()); c.qvals.assign((size_t)nq, 0.0); }
  for (uint64_t i = 0; i < nq && a.ok; i++) {
    a.object_preamble(K_QPAIR);
    a.str(c.qkeys[(size_t)i]);
    a.prim(c.qvals[(size_t)i]);
  }
  a.str(c.info);
}

bool io_graph(Archive& a, std::vector<NodeRec>& g) {
  std::string sig = "serialization::archive";
  a.str(sig);
  if (!a.saving && sig != "serialization::archive") return false;
  uint16_t ver = 17;                                    // BOOST_ARCHIVE_VERSION of Boost 1.71
  a.prim(ver);
  // binary_oarchive_impl::init() then calls basic_binary_oprimitive::init() (boost/archive/impl/basic_binary_oprimitive.ipp):
  // the native sizes of int, long, float, double as single bytes and int(1) as the endianness marker; basic_binary_iprimitive::
  // init() throws incompatible_native_format on any mismatch, so a reader must find exactly these 8 bytes (x86-64 Linux)
  uint8_t native[4] = {4, 8, 4, 8};
  for (int k = 0; k < 4; k++) a.prim(native[k]);
  int32_t endian = 1;
  a.prim(endian);
  if (!a.saving && (native[0] != 4 || native[1] != 8 || native[2] != 4 || native[3] != 8 || endian != 1)) return false;
  a.object_preamble(K_GRAPH);
  uint64_t n = g.size();
  io_count(a, n, K_PAIR);
  if (!a.saving) { if (n > (1u << 24) || !a.fits(n, 240)) return false; g.assign((size_t)n, NodeRec()); }   // a node: two poses (2 x 56) + motion (128) at the very least; a pose-only node is 273 bytes
  for (uint64_t i = 0; i < n && a.ok; i++) {
    a.object_preamble(K_PAIR);
    io_node(a, g[(size_t)i]);
    a.object_preamble(K_VCONS);
    uint64_t nc = g[(size_t)i].constraints.size();
    io_count(a, nc, K_CONS);
    if (!a.saving) { if (nc > (1u << 20) || !a.fits(nc, 300)) return false; g[(size_t)i].constraints.assign((size_t)nc, ConstraintRec()); }
    for (uint64_t j = 0; j < nc && a.ok; j++) io_constraint(a, g[(size_t)i].constraints[(size_t)j]);
  }
  return a.ok;
}

void cloud_in(Cloud& c, const cfear_graph_cloud& v) {
  c.present = v.n >= 0;
  c.src = v.n > 0 ? v.xyzi : nullptr;
  if (v.n > 0) c.xyzi.assign(v.xyzi, v.xyzi + (size_t)v.n * 4);
  c.stamp = v.stamp; c.seq = v.seq; c.frame_id = v.frame_id ? v.frame_id : "";
}
void cloud_out(const Cloud& c, cfear_graph_cloud* v) {
  v->xyzi = c.xyzi.empty() ? nullptr : c.xyzi.data();
  v->n = c.present ? (int32_t)(c.xyzi.size() / 4) : -1;
  v->stamp = c.stamp; v->seq = c.seq; v->frame_id = c.frame_id.c_str();
}

}  // namespace

struct cfear_graph { std::vector<NodeRec> nodes; };

extern "C" int cfear_graph_save(const char* path, const cfear_graph_node* nodes, int32_t n_nodes) {
  if (!path || n_nodes < 0 || (n_nodes > 0 && !nodes)) return CFEAR_ERR_INVALID_ARGUMENT;
  std::vector<NodeRec> g((size_t)n_nodes);
  for (int i = 0; i < n_nodes; i++) {
    const cfear_graph_node& s = nodes[i];
    NodeRec& d = g[(size_t)i];
    d.T = s.T; d.Tgt = s.Tgt; d.has_Tgt = s.has_Tgt != 0; d.idx = s.idx; d.stamp = s.stamp;
    memcpy(d.motion, s.motion, sizeof(d.motion));
    cloud_in(d.peaks, s.cloud_peaks); cloud_in(d.nopeaks, s.cloud_nopeaks); cloud_in(d.input, s.normal_input);
    d.has_normal = s.has_normal != 0; d.input_is_nopeaks = s.input_is_nopeaks != 0;
    if (s.n_cells > 0) { if (!s.cells) return CFEAR_ERR_INVALID_ARGUMENT; d.cells.assign(s.cells, s.cells + s.n_cells); }
    d.radius = s.radius; d.weight_intensity = s.weight_intensity != 0;
    for (int j = 0; j < s.n_constraints; j++) {
      const cfear_graph_constraint& c = s.constraints[j];
      ConstraintRec r;
      r.id_begin = c.id_begin; r.id_end = c.id_end; r.t_be = c.t_be; memcpy(r.information, c.information, sizeof(r.information));
      r.type = c.type;
      for (int q = 0; q < c.n_quality; q++) { r.qkeys.push_back(c.quality_keys[q]); r.qvals.push_back(c.quality_values[q]); }
      r.info = c.info ? c.info : "";
      d.constraints.push_back(r);
    }
  }
  Archive a;
  a.saving = true;
  a.f = fopen(path, "wb");
  if (!a.f) return CFEAR_ERR_IO;
  const bool ok = io_graph(a, g);
  const bool closed = fclose(a.f) == 0;
  return ok && closed ? CFEAR_OK : CFEAR_ERR_IO;
}

extern "C" int cfear_graph_load(const char* path, cfear_graph** out) {
  if (!path || !out) return CFEAR_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  Archive a;
  a.saving = false;
  a.f = fopen(path, "rb");
  if (!a.f) return CFEAR_ERR_IO;
  if (fseek(a.f, 0, SEEK_END) == 0) { const long sz = ftell(a.f); a.file_bytes = sz > 0 ? (uint64_t)sz : 0; }
  rewind(a.f);
  std::unique_ptr<cfear_graph> g(new cfear_graph());
  const bool ok = io_graph(a, g->nodes);
  fclose(a.f);
  if (!ok) return CFEAR_ERR_FORMAT;
  for (NodeRec& n : g->nodes) {
    n.constraint_views.resize(n.constraints.size());
    for (size_t j = 0; j < n.constraints.size(); j++) {
      ConstraintRec& r = n.constraints[j];
      r.qkey_ptrs.clear();
      for (const std::string& k : r.qkeys) r.qkey_ptrs.push_back(k.c_str());
      cfear_graph_constraint& v = n.constraint_views[j];
      v.id_begin = r.id_begin; v.id_end = r.id_end; v.t_be = r.t_be; memcpy(v.information, r.information, sizeof(v.information));
      v.type = r.type; v.n_quality = (int32_t)r.qkeys.size();
      v.quality_keys = r.qkey_ptrs.empty() ? nullptr : r.qkey_ptrs.data();
      v.quality_values = r.qvals.empty() ? nullptr : r.qvals.data();
      v.info = r.info.c_str();
    }
  }
  *out = g.release();
  return CFEAR_OK;
}

extern "C" int cfear_graph_size(const cfear_graph* g) { return g ? (int)g->nodes.size() : CFEAR_ERR_INVALID_ARGUMENT; }

extern "C" int cfear_graph_node_at(const cfear_graph* g, int32_t i, cfear_graph_node* out) {
  if (!g || !out || i < 0 || i >= (int32_t)g->nodes.size()) return CFEAR_ERR_INVALID_ARGUMENT;
  const NodeRec& n = g->nodes[(size_t)i];
  memset(out, 0, sizeof(*out));
  out->T = n.T; out->Tgt = n.Tgt; out->has_Tgt = n.has_Tgt; out->idx = n.idx; out->stamp = n.stamp;
  memcpy(out->motion, n.motion, sizeof(out->motion));
  cloud_out(n.peaks, &out->cloud_peaks); cloud_out(n.nopeaks, &out->cloud_nopeaks); cloud_out(n.input, &out->normal_input);
  out->has_normal = n.has_normal; out->input_is_nopeaks = n.input_is_nopeaks;
  out->cells = n.cells.empty() ? nullptr : n.cells.data();
  out->n_cells = (int32_t)n.cells.size();
  out->radius = n.radius; out->weight_intensity = n.weight_intensity;
  out->constraints = n.constraint_views.empty() ? nullptr : n.constraint_views.data();
  out->n_constraints = (int32_t)n.constraint_views.size();
  return CFEAR_OK;
}

extern "C" int cfear_graph_destroy(cfear_graph* g) {
  delete g;
  return CFEAR_OK;
}

// ---- pose conversions of the graph types (types.cpp:5-11, 25-43) ---------------------------------------------------
// PoseEigToCeres for a planar pose: Quaterniond(R_z(theta)).normalize() = (0, 0, sin(theta/2), cos(theta/2)) with w >= 0
// (Eigen's rotation-matrix constructor takes the positive root of the trace branch for |theta| < 2 pi / 3 and the z
// branch beyond; both give the quaternion with w >= 0 for theta in (-pi, pi]).
extern "C" void cfear_pose3d_from_xyt(const double xyt[3], cfear_pose3d* out) {
  out->p[0] = xyt[0]; out->p[1] = xyt[1]; out->p[2] = 0.0;
  const double h = 0.5 * std::atan2(std::sin(xyt[2]), std::cos(xyt[2]));
  out->q[0] = 0.0; out->q[1] = 0.0; out->q[2] = std::sin(h); out->q[3] = std::cos(h);
}
extern "C" void cfear_pose3d_to_xyt(const cfear_pose3d* p, double xyt[3]) {
  xyt[0] = p->p[0]; xyt[1] = p->p[1];
  const double x = p->q[0], y = p->q[1], z = p->q[2], w = p->q[3];
  xyt[2] = std::atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z));     // Affine3dToVectorXYeZ of PoseCeresToEig
}
