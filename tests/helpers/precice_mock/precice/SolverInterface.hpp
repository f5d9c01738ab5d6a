// TEST-ONLY stand-in for preCICE's pre-1.0 precice/SolverInterface.hpp (the API the reference adapter uses,
// fem-shell_precice.cpp:15, 50-52, 72-170, 256-412), so that fem-shell_amd/host/coupling.cpp can be syntax-checked with
// -DFEMSHELL_HAVE_PRECICE in an image without preCICE.  NOT preCICE and NOT a reference build: declarations only,
// nothing is linked or run.
#pragma once
#include <string>
namespace precice {
class SolverInterface {
  public:
    SolverInterface(const std::string &participant, const std::string &config, int rank, int size);
    int getDimensions() const;
    int getMeshID(const std::string &name) const;
    int getDataID(const std::string &name, int meshID) const;
    void setMeshVertices(int meshID, int size, const double *positions, int *ids);
    double initialize();
    bool isActionRequired(const std::string &action) const;
    void fulfilledAction(const std::string &action);
    void initializeData();
    bool isReadDataAvailable() const;
    void writeBlockVectorData(int dataID, int size, const int *ids, const double *values);
    void readBlockVectorData(int dataID, int size, const int *ids, double *values) const;
    double advance(double dt);
    bool isCouplingOngoing() const;
    void finalize();
};
} // namespace precice
