bash tools/r06_probes/gnb_tests.sh
bash tools/r06_probes/gnb_trace.sh
