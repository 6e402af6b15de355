/* arbplf-dwell: JSON on stdin -> JSON on stdout, exit status 0 on success.
 * Drop-in for the reference's src/arbplf-dwell.c (run_json_script). */
#include "arbplf.h"

int main(void)
{
    return arbplf_run_stdin(arbplf_dwell_string);
}
