#include "boost/iostreams/filtering_stream.hpp"
