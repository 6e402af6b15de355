/* arbplf-ll: JSON on stdin -> JSON on stdout, exit status 0 on success.
 * Drop-in for the reference's src/arbplf-ll.c:4-15 (run_json_script). */
#include "arbplf.h"

int main(void)
{
    return arbplf_run_stdin(arbplf_ll_string);
}
