// bramble command line: see br_cli_main (include/bramble_amd.h)
extern "C" int br_cli_main(int argc, char **argv);
int main(int argc, char **argv) { return br_cli_main(argc, argv); }
