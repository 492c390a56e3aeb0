// Build configuration (the reference generates this file from config.in with CMake;
// here the two switches are plain -D flags: -DCUDDH_DEBUG, -DCUDDH_LOG_MEMCPY).
#ifndef CUDDH_CONFIG_HPP
#define CUDDH_CONFIG_HPP
#endif
