#pragma once
#include <embedding/embedding.hpp>  /* oracle/ref_shims/ebc: all declarations live there */
