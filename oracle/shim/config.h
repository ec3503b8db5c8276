/*
 * oracle/shim/config.h -- TEST INFRASTRUCTURE ONLY.
 * Stand-in for the autoconf-generated config.h of the reference build
 * (/root/reference/configure.ac:19-76): file backend + benchmark backend only.
 */
#ifndef ORACLE_SHIM_CONFIG_H
#define ORACLE_SHIM_CONFIG_H
#define VERSION "0.24-oracle"
#define PACKAGE_STRING "minimodem 0.24-oracle"
#define USE_SNDFILE 1
#define USE_BENCHMARKS 1
#define USE_ALSA 0
#define USE_PULSEAUDIO 0
#define USE_SNDIO 0
#endif
