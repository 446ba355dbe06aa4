/* placeholder translation unit; the examples/cpu.rs restatement lands here next. */
int tmo_cpu_path_present(void) { return 0; }
