// plugins/customized_measurement.so for Mitsuba 0.6 (README.md:1: "customized_measurment brdf pluggin")
#include "measured_bsdf.hpp"

MTS_NAMESPACE_BEGIN
MTS_IMPLEMENT_CLASS_S(CustomizedMeasurement, false, BSDF)
MTS_NAMESPACE_END
MTS_EXPORT_PLUGIN(CustomizedMeasurement, "Customized measured BRDF table (MI355X / libmerl_hip)")
