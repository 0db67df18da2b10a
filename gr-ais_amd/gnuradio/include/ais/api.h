/* Export macro of the gr::ais blocks built over libaisx.so (role of the reference's include/ais/api.h). */
#ifndef AISX_GR_AIS_API_H
#define AISX_GR_AIS_API_H

#include <gnuradio/attributes.h>

#if defined(gnuradio_ais_EXPORTS) || defined(gnuradio_ais_amd_EXPORTS)
#define AIS_API __GR_ATTR_EXPORT
#else
#define AIS_API __GR_ATTR_IMPORT
#endif

#endif
