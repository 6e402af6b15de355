/* arbplf-hess: JSON on stdin -> JSON on stdout, exit status 0 on success.
 * Drop-in for the reference's src/arbplf-hess.c (run_json_script with hess_query). */
#include "arbplf.h"

int main(void)
{
    return arbplf_run_stdin(arbplf_hess_string);
}
