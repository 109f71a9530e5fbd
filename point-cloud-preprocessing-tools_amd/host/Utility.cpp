#include "Utility.h"

#include <iostream>

#include "../../include/bev_mi355x.h"

static const char *sensor_name(SensorType t)
{
    switch (t) {
    case HDL_32E: return "HDL_32E";
    case HDL_64E: return "HDL_64E";
    case OS1_64: return "OS1_64";
    default: return nullptr;
    }
}

/* substring match in the reference's order of tests (src/Utility.cpp:74-83) */
SensorType parseSensorType(std::string sensor_str)
{
    for (SensorType t : {HDL_32E, HDL_64E, OS1_64})
        if (sensor_str.find(sensor_name(t)) != std::string::npos) return t;
    std::cerr << "Unknown sensor type: " << sensor_str << "!" << std::endl;
    return UNKNOWN;
}

/* the table itself lives behind the C ABI (bev_params_for_sensor) so that the
 * kernels and the host can never disagree about it */
SensorParams getSensorParams(SensorType sensor_type)
{
    SensorParams sp{0, 0, 0, 0.0f};
    bev_params_t bp;
    const char *name = sensor_name(sensor_type);
    if (!name || bev_params_for_sensor(name, &bp) != BEV_OK) {
        std::cerr << "Unknown sensor type! " << std::endl;
        return sp;
    }
    sp.N_SCAN = bp.n_scan;
    sp.Horizon_SCAN = bp.horizon_scan;
    sp.GROUND_UPPER_SCAN = bp.ground_upper_scan;
    sp.HEIGHT_RES = bp.height_res;
    return sp;
}

std::string printSensorParams(SensorParams params)
{
    return "N_SCAN: " + std::to_string(params.N_SCAN) + ", Horizon_SCAN: " + std::to_string(params.Horizon_SCAN) +
           ", GROUND_UPPER_SCAN: " + std::to_string(params.GROUND_UPPER_SCAN);
}
