// rdvio_yaml.hpp -- the YAML the reference reads, without yaml-cpp (not available in this build image).
//
//  * a small reader for the subset of YAML that configs/setting.yaml, configs/euroc_sensor.yaml and EuRoC's
//    mav0/<sensor>/sensor.yaml use: block mappings nested by indentation, flow sequences `[a, b, ...]` (may span lines),
//    plain / quoted scalars, `#` comments, the OpenCV-style `%YAML:1.0` directive line;
//  * load_yaml_config(): key for key what rdvio::extra::YamlConfig::YamlConfig does
//    (/root/reference/src/rdvio_extra/src/yaml_config.cpp:83-338): defaults from rdvio::Config
//    (src/rdvio/src/config.cpp), mandatory cam0.* / imu.* keys of the device file, optional groups of the SLAM file, the
//    same four exception kinds with the same messages (src/rdvio_extra/include/rdvio/extra/yaml_config.h:10-27).
//    One member has no default in the reference: m_parsac_keyframe_check_size is not initialised in the constructor
//    (yaml_config.cpp:100-118 lists the other parsac members only), so a SLAM file without parsac.keyframe_check_size leaves
//    it indeterminate there; here it keeps rdvio::Config's value (3, config.cpp:71).
//    Keys the hot path does not consume (cam0.distortion, camera_distortion_flag, time_offset, initializer.refine_imu,
//    solver.time_limit, parsac.dynamic_probability / threshold / norm_scale) are still type-checked like the reference
//    does and returned in YamlExtras.
#pragma once

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rdvio_pipeline.h"

namespace rdvio_hip {

struct YamlException : std::runtime_error {
    explicit YamlException(const std::string &what) : std::runtime_error(what) {}
};
struct YamlLoadException : YamlException {
    explicit YamlLoadException(const std::string &filename) : YamlException("cannot load config " + filename) {}
};
struct YamlParseException : YamlException {
    explicit YamlParseException(const std::string &message) : YamlException(message) {}
};
struct YamlConfigMissingException : YamlException {
    explicit YamlConfigMissingException(const std::string &path) : YamlException("config \"" + path + "\" is mandatory") {}
};
struct YamlTypeErrorException : YamlException {
    explicit YamlTypeErrorException(const std::string &path) : YamlException("config \"" + path + "\" has wrong type") {}
};

struct YamlNode {
    enum Kind { Null, Scalar, Sequence, Map } kind = Null;
    std::string scalar;
    std::vector<YamlNode> seq;
    std::vector<std::pair<std::string, YamlNode>> map;
    const YamlNode *find(const std::string &key) const {
        for (const auto &kv : map)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    // dotted path, like find_node (yaml_config.cpp:7-23)
    const YamlNode *path(const std::string &dotted) const {
        const YamlNode *n = this;
        std::stringstream ss(dotted);
        std::string child;
        while (n && std::getline(ss, child, '.')) n = n->kind == Map ? n->find(child) : nullptr;
        return (n && n->kind != Null) ? n : nullptr;
    }
};

namespace yaml_detail {

inline std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}
// strip a `#` comment that is not inside quotes
inline std::string strip_comment(const std::string &s) {
    char quote = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (quote) {
            if (c == quote) quote = 0;
        } else if (c == '"' || c == '\'') {
            quote = c;
        } else if (c == '#' && (i == 0 || std::isspace((unsigned char)s[i - 1]))) {
            return s.substr(0, i);
        }
    }
    return s;
}
inline std::string unquote(const std::string &s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}
inline YamlNode parse_flow_sequence(const std::string &text, size_t line_no) {
    // text starts with '[' and contains the matching ']' (nested flow sequences are not used by the reference's files)
    YamlNode n;
    n.kind = YamlNode::Sequence;
    const size_t close = text.rfind(']');
    if (text.empty() || text[0] != '[' || close == std::string::npos) throw YamlParseException("yaml: malformed flow sequence near line " + std::to_string(line_no));
    std::string item;
    std::stringstream ss(text.substr(1, close - 1));
    while (std::getline(ss, item, ',')) {
        const std::string v = trim(item);
        if (v.empty()) continue;
        YamlNode e;
        e.kind = YamlNode::Scalar;
        e.scalar = unquote(v);
        n.seq.push_back(e);
    }
    return n;
}

struct Line {
    int indent;
    std::string text;
    size_t no;
};

inline YamlNode parse_block(const std::vector<Line> &lines, size_t &pos, int indent) {
    YamlNode node;
    node.kind = YamlNode::Map;
    while (pos < lines.size()) {
        const Line &ln = lines[pos];
        if (ln.indent < indent) break;
        if (ln.indent > indent) throw YamlParseException("yaml: unexpected indentation at line " + std::to_string(ln.no));
        const size_t colon = ln.text.find(':');
        if (colon == std::string::npos || ln.text[0] == '-') throw YamlParseException("yaml: expected `key: value` at line " + std::to_string(ln.no));
        const std::string key = unquote(trim(ln.text.substr(0, colon)));
        std::string rest = trim(ln.text.substr(colon + 1));
        ++pos;
        YamlNode value;
        if (rest.empty()) {
            if (pos < lines.size() && lines[pos].indent > indent) {
                if (!lines[pos].text.empty() && lines[pos].text[0] == '[') {
                    // a flow sequence that starts on the next line
                    std::string acc;
                    const size_t first = lines[pos].no;
                    while (pos < lines.size() && lines[pos].indent > indent) {
                        acc += lines[pos].text + " ";
                        const bool done = lines[pos].text.find(']') != std::string::npos;
                        ++pos;
                        if (done) break;
                    }
                    value = parse_flow_sequence(trim(acc), first);
                } else {
                    value = parse_block(lines, pos, lines[pos].indent);
                }
            }
        } else if (rest[0] == '[') {
            const size_t first = ln.no;
            while (rest.find(']') == std::string::npos) {  // continues on the following lines
                if (pos >= lines.size()) throw YamlParseException("yaml: unterminated flow sequence at line " + std::to_string(first));
                rest += " " + lines[pos].text;
                ++pos;
            }
            value = parse_flow_sequence(rest, first);
        } else {
            value.kind = YamlNode::Scalar;
            value.scalar = unquote(rest);
        }
        node.map.emplace_back(key, value);
    }
    return node;
}

}  // namespace yaml_detail

inline YamlNode parse_yaml(const std::string &text) {
    using namespace yaml_detail;
    std::vector<Line> lines;
    std::stringstream ss(text);
    std::string raw;
    size_t no = 0;
    while (std::getline(ss, raw)) {
        ++no;
        if (!raw.empty() && raw.back() == '\r') raw.pop_back();
        if (raw.rfind("%YAML", 0) == 0 || raw.rfind("---", 0) == 0) continue;
        const std::string body = strip_comment(raw);
        if (trim(body).empty()) continue;
        int indent = 0;
        while ((size_t)indent < body.size() && body[(size_t)indent] == ' ') ++indent;
        if ((size_t)indent < body.size() && body[(size_t)indent] == '\t') throw YamlParseException("yaml: tab indentation at line " + std::to_string(no));
        lines.push_back({indent, trim(body), no});
    }
    size_t pos = 0;
    YamlNode root = parse_block(lines, pos, lines.empty() ? 0 : lines[0].indent);
    if (pos != lines.size()) throw YamlParseException("yaml: unexpected indentation at line " + std::to_string(lines[pos].no));
    return root;
}

inline YamlNode load_yaml_file(const std::string &filename) {
    std::ifstream f(filename);
    if (!f) throw YamlLoadException(filename);
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_yaml(ss.str());
}

// values of keys the hot path does not consume (parsed and type-checked like the reference does)
struct YamlExtras {
    double camera_distortion[4] = {0, 0, 0, 0};
    int camera_distortion_flag = 0;
    double camera_time_offset = 0.0;
    int initializer_refine_imu = 1;
    double solver_time_limit = 1.0e6;
    double parsac_dynamic_probability = 0.0, parsac_threshold = 3.0, parsac_norm_scale = 1.0;
};

namespace yaml_detail {

inline double as_double(const YamlNode &n, const std::string &path) {
    if (n.kind != YamlNode::Scalar) throw YamlTypeErrorException(path);
    char *end = nullptr;
    const double v = std::strtod(n.scalar.c_str(), &end);
    if (end == n.scalar.c_str() || *end != '\0') throw YamlTypeErrorException(path);
    return v;
}
inline long as_size(const YamlNode &n, const std::string &path) {
    if (n.kind != YamlNode::Scalar) throw YamlTypeErrorException(path);
    char *end = nullptr;
    const long v = std::strtol(n.scalar.c_str(), &end, 10);
    if (end == n.scalar.c_str() || *end != '\0' || v < 0) throw YamlTypeErrorException(path);
    return v;
}
inline bool as_bool(const YamlNode &n, const std::string &path) {
    if (n.kind != YamlNode::Scalar) throw YamlTypeErrorException(path);
    std::string s = n.scalar;
    for (char &c : s) c = (char)std::tolower((unsigned char)c);
    if (s == "true" || s == "yes" || s == "on" || s == "y") return true;   // yaml-cpp's bool spellings
    if (s == "false" || s == "no" || s == "off" || s == "n") return false;
    throw YamlTypeErrorException(path);
}
inline void as_vector(const YamlNode &n, const std::string &path, double *out, size_t count) {
    if (n.kind != YamlNode::Sequence || n.seq.size() != count) throw YamlTypeErrorException(path);  // require_vector
    for (size_t i = 0; i < count; ++i) out[i] = as_double(n.seq[i], path);
}

}  // namespace yaml_detail

// rdvio::extra::YamlConfig(slam_config_filename, device_config_filename) -> the pipeline's config struct
inline rdvio_pipeline_config load_yaml_config(const std::string &slam_config_filename, const std::string &device_config_filename,
                                              YamlExtras *extras = nullptr) {
    using namespace yaml_detail;
    rdvio_pipeline_config c;
    rdvio_pipeline_config_default(&c);   // rdvio::Config defaults (config.cpp)
    YamlExtras ex;
    const YamlNode slam = load_yaml_file(slam_config_filename);
    const YamlNode dev = load_yaml_file(device_config_filename);
    auto need = [&](const char *path) -> const YamlNode & {
        const YamlNode *n = dev.path(path);
        if (!n) throw YamlConfigMissingException(path);
        return *n;
    };
    // ---- device file: every key is mandatory (yaml_config.cpp:141-211)
    {
        double v4[4];
        as_vector(need("cam0.intrinsics"), "cam0.intrinsics", v4, 4);
        for (double &k : c.K) k = 0.0;
        c.K[0] = v4[0]; c.K[4] = v4[1]; c.K[2] = v4[2]; c.K[5] = v4[3]; c.K[8] = 1.0;
        as_vector(need("cam0.distortion"), "cam0.distortion", ex.camera_distortion, 4);
        ex.camera_distortion_flag = (int)as_size(need("cam0.camera_distortion_flag"), "cam0.camera_distortion_flag");
        ex.camera_time_offset = as_double(need("cam0.time_offset"), "cam0.time_offset");
        double res[2];
        as_vector(need("cam0.resolution"), "cam0.resolution", res, 2);
        c.width = (int32_t)res[0];
        c.height = (int32_t)res[1];
        as_vector(need("cam0.extrinsic.q_bc"), "cam0.extrinsic.q_bc", c.q_bc, 4);
        as_vector(need("cam0.extrinsic.p_bc"), "cam0.extrinsic.p_bc", c.p_bc, 3);
        as_vector(need("cam0.noise"), "cam0.noise", c.keypoint_noise_cov, 4);
        as_vector(need("imu.extrinsic.q_bi"), "imu.extrinsic.q_bi", c.q_bi, 4);
        as_vector(need("imu.extrinsic.p_bi"), "imu.extrinsic.p_bi", c.p_bi, 3);
        as_vector(need("imu.noise.cov_g"), "imu.noise.cov_g", c.gyroscope_noise_cov, 9);
        as_vector(need("imu.noise.cov_a"), "imu.noise.cov_a", c.accelerometer_noise_cov, 9);
        as_vector(need("imu.noise.cov_bg"), "imu.noise.cov_bg", c.gyroscope_bias_noise_cov, 9);
        as_vector(need("imu.noise.cov_ba"), "imu.noise.cov_ba", c.accelerometer_bias_noise_cov, 9);
    }
    // ---- SLAM file: every key optional (yaml_config.cpp:213-338)
    auto opt = [&](const char *path) { return slam.path(path); };
    if (auto n = opt("output.q_bo")) as_vector(*n, "output.q_bo", c.q_bo, 4);
    if (auto n = opt("output.p_bo")) as_vector(*n, "output.p_bo", c.p_bo, 3);
    if (auto n = opt("sliding_window.size")) c.sliding_window_size = (int32_t)as_size(*n, "sliding_window.size");
    if (auto n = opt("sliding_window.subframe_size")) c.sliding_window_subframe_size = (int32_t)as_size(*n, "sliding_window.subframe_size");
    if (auto n = opt("sliding_window.tracker_frequent")) c.sliding_window_tracker_frequent = (int32_t)as_size(*n, "sliding_window.tracker_frequent");
    if (auto n = opt("sliding_window.force_keyframe_landmarks"))
        c.sliding_window_force_keyframe_landmarks = (int32_t)as_size(*n, "sliding_window.force_keyframe_landmarks");
    if (auto n = opt("feature_tracker.min_keypoint_distance")) c.feature_tracker_min_keypoint_distance = as_double(*n, "feature_tracker.min_keypoint_distance");
    if (auto n = opt("feature_tracker.max_keypoint_detection"))
        c.feature_tracker_max_keypoint_detection = (int32_t)as_size(*n, "feature_tracker.max_keypoint_detection");
    if (auto n = opt("feature_tracker.max_init_frames")) c.feature_tracker_max_init_frames = (int32_t)as_size(*n, "feature_tracker.max_init_frames");
    if (auto n = opt("feature_tracker.max_frames")) c.feature_tracker_max_frames = (int32_t)as_size(*n, "feature_tracker.max_frames");
    if (auto n = opt("feature_tracker.clahe_clip_limit")) c.feature_tracker_clahe_clip_limit = as_double(*n, "feature_tracker.clahe_clip_limit");
    if (auto n = opt("feature_tracker.clahe_width")) c.feature_tracker_clahe_width = (int32_t)as_size(*n, "feature_tracker.clahe_width");
    if (auto n = opt("feature_tracker.clahe_height")) c.feature_tracker_clahe_height = (int32_t)as_size(*n, "feature_tracker.clahe_height");
    if (auto n = opt("feature_tracker.predict_keypoints")) c.feature_tracker_predict_keypoints = as_bool(*n, "feature_tracker.predict_keypoints") ? 1 : 0;
    if (auto n = opt("initializer.keyframe_num")) c.initializer_keyframe_num = (int32_t)as_size(*n, "initializer.keyframe_num");
    if (auto n = opt("initializer.keyframe_gap")) c.initializer_keyframe_gap = (int32_t)as_size(*n, "initializer.keyframe_gap");
    if (auto n = opt("initializer.min_matches")) c.initializer_min_matches = (int32_t)as_size(*n, "initializer.min_matches");
    if (auto n = opt("initializer.min_parallax")) c.initializer_min_parallax = as_double(*n, "initializer.min_parallax");
    if (auto n = opt("initializer.min_triangulation")) c.initializer_min_triangulation = (int32_t)as_size(*n, "initializer.min_triangulation");
    if (auto n = opt("initializer.min_landmarks")) c.initializer_min_landmarks = (int32_t)as_size(*n, "initializer.min_landmarks");
    if (auto n = opt("initializer.refine_imu")) ex.initializer_refine_imu = c.initializer_refine_imu = as_bool(*n, "initializer.refine_imu") ? 1 : 0;
    if (auto n = opt("solver.iteration_limit")) c.solver_iteration_limit = (int32_t)as_size(*n, "solver.iteration_limit");
    if (auto n = opt("solver.time_limit")) {
        ex.solver_time_limit = as_double(*n, "solver.time_limit");
        // Ceres' max_solver_time_in_seconds (solver.cpp:188).  The device solver is bounded by solver.iteration_limit only: a limit
        // that could cut a solve short (the shipped files say 1e6 s) is refused rather than silently ignored
        if (ex.solver_time_limit < 10.0)
            throw YamlParseException("config \"solver.time_limit\" below 10 s is not supported: the device solver has no wall-clock limit, bound it with solver.iteration_limit");
    }
    if (auto n = opt("parsac.parsac_flag")) c.parsac_flag = as_bool(*n, "parsac.parsac_flag") ? 1 : 0;
    if (auto n = opt("parsac.dynamic_probability")) ex.parsac_dynamic_probability = as_double(*n, "parsac.dynamic_probability");
    if (auto n = opt("parsac.threshold")) ex.parsac_threshold = as_double(*n, "parsac.threshold");
    if (auto n = opt("parsac.norm_scale")) ex.parsac_norm_scale = as_double(*n, "parsac.norm_scale");
    if (auto n = opt("parsac.keyframe_check_size")) c.parsac_keyframe_check_size = (int32_t)as_size(*n, "parsac.keyframe_check_size");
    if (auto n = opt("rotation.misalignment_threshold")) c.rotation_misalignment_threshold = as_double(*n, "rotation.misalignment_threshold");
    if (auto n = opt("rotation.ransac_threshold")) c.rotation_ransac_threshold = as_double(*n, "rotation.ransac_threshold");
    if (extras) *extras = ex;
    return c;
}

}  // namespace rdvio_hip
