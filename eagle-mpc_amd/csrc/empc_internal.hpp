// internal helpers shared by the translation units of libempc.so
#pragma once
#include <string>
namespace empc {
void set_last_error(const std::string& msg);
}
