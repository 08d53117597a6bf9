/* placeholder translation unit; classical restatements are added in a later commit */
int orc_classical_placeholder(void) { return 0; }
