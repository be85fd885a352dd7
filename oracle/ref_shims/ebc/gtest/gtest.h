/* TEST INFRASTRUCTURE ONLY -- the two gtest macros the reference's CPU embedding_collection
 * reference uses, as throwing checks (oracle/ref_ebc_shim.cpp). */
#pragma once
#include <stdexcept>
#define ASSERT_TRUE(cond) \
  do { if (!(cond)) throw std::runtime_error("ASSERT_TRUE failed: " #cond); } while (0)
#define ASSERT_EQ(a, b) \
  do { if (!((a) == (b))) throw std::runtime_error("ASSERT_EQ failed: " #a " == " #b); } while (0)
