/*
 * Utility.h — sensor table of the hot path (mirror of the reference's
 * include/Utility.h:22-36,82-84 and src/Utility.cpp:72-124).  Same names and
 * argument meaning; the Eigen-based pose helpers of the reference's Utility are
 * part of the label step ("next" row N1) and live in LabelStep.h.
 */
#ifndef BEV_HOST_UTILITY_H
#define BEV_HOST_UTILITY_H

#include <string>

enum SensorType { HDL_32E = 0, HDL_64E, OS1_64, UNKNOWN };

struct SensorParams {
    int Horizon_SCAN;
    int N_SCAN;
    int GROUND_UPPER_SCAN;
    float HEIGHT_RES;
};

SensorType parseSensorType(std::string sensor_str);
/* Unknown type: the reference returns an uninitialised struct (src/Utility.cpp:119-123);
 * here every field is 0 and an error is printed, and callers must check N_SCAN > 0. */
SensorParams getSensorParams(SensorType sensor_type);
std::string printSensorParams(SensorParams params);

#endif
