#include "mock_libmesh.h" // test-only stand-in, see mock_libmesh.h
