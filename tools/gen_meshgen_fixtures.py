"""Writes tests/golden/meshgen_ref/: output files of the REFERENCE's meshGen (compiled as-is from
/root/reference/src/meshgen/main_all.cpp into oracle/_ref/meshGen_ref by `make -C oracle ref`; std-only source, no
stand-ins) for the argument sets of CASES.  Run in the build container, where /root/reference exists; the files are
data fixtures the twin (fem-shell_amd/host/meshGen) is byte-compared with on any machine."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "meshGen_ref")
OUT = os.path.join(ROOT, "tests", "golden", "meshgen_ref")

CASES = {
    "tri_z_uniform": "t 7 5 -1.5 0 2 3 0,1,-1,2 2.5 2 1 z",
    "tri_y_flap_unit": "t 4 9 0 0 0.1 1 2,20,2,2 1 1 0 y",
    "quad_z_testD16": "q 16 16 0 0 10 10 0,0,0,0 300 2 1 z",
    "quad_x_testF_con": "q 3 3 0 0 10 2 1,1,1,1 0.0004 1 1 x",
    "quad_z_noload": "q 5 2 -3 1 7 8.5 -1,21,-1,-1 1e-4 0 1 z",
    "tri_z_thirds": "t 3 3 0 0 1 1 0,0,0,0 1 2 1 z",
}

if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "CASES.txt"), "w") as f:
        for name, args in CASES.items():
            subprocess.check_call([REF] + args.split() + [os.path.join(OUT, name)], stdout=subprocess.DEVNULL)
            f.write("%s: %s\n" % (name, args))
    print("wrote", OUT)
