// prints the rdvio_pipeline_config that load_yaml_config(config, calib) produces as one JSON object, or
// "EXCEPTION <kind>: <message>" (tests/test_euroc_harness.py compares with the Python mapping and checks the messages
// against /root/reference/src/rdvio_extra/include/rdvio/extra/yaml_config.h:10-27)
#include <cstdio>

#include "../../rd_vio_amd/host/rdvio_yaml.hpp"

static void arr(const char *name, const double *v, int n, bool last = false) {
    std::printf("\"%s\": [", name);
    for (int i = 0; i < n; ++i) std::printf(i ? ", %.17g" : "%.17g", v[i]);
    std::printf(last ? "]" : "], ");
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    using namespace rdvio_hip;
    try {
        YamlExtras ex;
        const rdvio_pipeline_config c = load_yaml_config(argv[1], argv[2], &ex);
        std::printf("{\"width\": %d, \"height\": %d, ", c.width, c.height);
        arr("K", c.K, 9);
        arr("q_bc", c.q_bc, 4); arr("p_bc", c.p_bc, 3); arr("q_bi", c.q_bi, 4); arr("p_bi", c.p_bi, 3); arr("q_bo", c.q_bo, 4); arr("p_bo", c.p_bo, 3);
        arr("keypoint_noise_cov", c.keypoint_noise_cov, 4);
        arr("gyroscope_noise_cov", c.gyroscope_noise_cov, 9); arr("accelerometer_noise_cov", c.accelerometer_noise_cov, 9);
        arr("gyroscope_bias_noise_cov", c.gyroscope_bias_noise_cov, 9); arr("accelerometer_bias_noise_cov", c.accelerometer_bias_noise_cov, 9);
        std::printf("\"sliding_window_size\": %d, \"sliding_window_subframe_size\": %d, \"sliding_window_force_keyframe_landmarks\": %d, "
                    "\"sliding_window_tracker_frequent\": %d, \"feature_tracker_min_keypoint_distance\": %.17g, "
                    "\"feature_tracker_max_keypoint_detection\": %d, \"feature_tracker_max_init_frames\": %d, \"feature_tracker_max_frames\": %d, "
                    "\"feature_tracker_clahe_clip_limit\": %.17g, \"feature_tracker_clahe_width\": %d, \"feature_tracker_clahe_height\": %d, "
                    "\"feature_tracker_predict_keypoints\": %d, \"initializer_keyframe_num\": %d, \"initializer_keyframe_gap\": %d, "
                    "\"initializer_min_matches\": %d, \"initializer_min_triangulation\": %d, \"initializer_min_landmarks\": %d, "
                    "\"initializer_min_parallax\": %.17g, \"solver_iteration_limit\": %d, \"rotation_misalignment_threshold\": %.17g, "
                    "\"rotation_ransac_threshold\": %.17g, \"random\": %d, \"parsac_flag\": %d, \"parsac_keyframe_check_size\": %d, \"initializer_refine_imu\": %d, ",
                    c.sliding_window_size, c.sliding_window_subframe_size, c.sliding_window_force_keyframe_landmarks, c.sliding_window_tracker_frequent,
                    c.feature_tracker_min_keypoint_distance, c.feature_tracker_max_keypoint_detection, c.feature_tracker_max_init_frames,
                    c.feature_tracker_max_frames, c.feature_tracker_clahe_clip_limit, c.feature_tracker_clahe_width, c.feature_tracker_clahe_height,
                    c.feature_tracker_predict_keypoints, c.initializer_keyframe_num, c.initializer_keyframe_gap, c.initializer_min_matches,
                    c.initializer_min_triangulation, c.initializer_min_landmarks, c.initializer_min_parallax, c.solver_iteration_limit,
                    c.rotation_misalignment_threshold, c.rotation_ransac_threshold, c.random, c.parsac_flag, c.parsac_keyframe_check_size, c.initializer_refine_imu);
        std::printf("\"extras\": {\"camera_distortion_flag\": %d, \"camera_time_offset\": %.17g, \"initializer_refine_imu\": %d, \"solver_time_limit\": %.17g, "
                    "\"parsac_dynamic_probability\": %.17g, \"parsac_threshold\": %.17g, \"parsac_norm_scale\": %.17g, ",
                    ex.camera_distortion_flag, ex.camera_time_offset, ex.initializer_refine_imu, ex.solver_time_limit, ex.parsac_dynamic_probability,
                    ex.parsac_threshold, ex.parsac_norm_scale);
        arr("camera_distortion", ex.camera_distortion, 4, true);
        std::printf("}}\n");
    } catch (const YamlLoadException &e) {
        std::printf("EXCEPTION load: %s\n", e.what());
    } catch (const YamlParseException &e) {
        std::printf("EXCEPTION parse: %s\n", e.what());
    } catch (const YamlConfigMissingException &e) {
        std::printf("EXCEPTION missing: %s\n", e.what());
    } catch (const YamlTypeErrorException &e) {
        std::printf("EXCEPTION type: %s\n", e.what());
    }
    return 0;
}
