// src/bsdfs/customized_measurement.cpp for Mitsuba 3 (README.md:1: "customized_measurment brdf pluggin")
#include "measured_bsdf.hpp"

NAMESPACE_BEGIN(mitsuba)
MI_IMPLEMENT_CLASS_VARIANT(CustomizedMeasurement, BSDF)
NAMESPACE_END(mitsuba)
MI_EXPORT_PLUGIN(CustomizedMeasurement, "Customized measured BRDF table (MI355X / libmerl_hip)")
