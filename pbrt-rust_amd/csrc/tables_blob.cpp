// tables_blob.cpp -- embeds data/sobol_tables.bin (DATA extracted from core/sobolmatrices.rs by
// tools/extract_sobol_tables.py) into the shared library, so the product needs no file at run time.
__asm__(
    ".section .rodata\n"
    ".global pt_sobol_blob\n"
    ".type pt_sobol_blob, @object\n"
    ".balign 16\n"
    "pt_sobol_blob:\n"
    ".incbin \"" PT_TABLES_PATH "\"\n"
    "pt_sobol_blob_end:\n"
    ".global pt_sobol_blob_size\n"
    ".type pt_sobol_blob_size, @object\n"
    ".balign 4\n"
    "pt_sobol_blob_size:\n"
    ".int pt_sobol_blob_end - pt_sobol_blob\n"
    ".section .note.GNU-stack,\"\",@progbits\n");
