/*
 * host_json.h -- minimal strict JSON reader / writer for the arbplf boundary.
 *
 * The reference parses with jansson (json_loads, flags 0; src/runjson.c:27) and
 * prints with json_dumps (src/runjson.c:46).  jansson is not a dependency here;
 * this reader accepts the same grammar (RFC 8259, top level must be an array or
 * an object, integers and reals are distinct types, duplicate keys: last wins)
 * and the writer formats reals the way jansson does (%.17g, always with a
 * fraction or exponent).  Values live in an arena freed by json_doc_free().
 */
#ifndef HOST_JSON_H
#define HOST_JSON_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { J_NULL, J_TRUE, J_FALSE, J_INT, J_REAL, J_STRING, J_ARRAY, J_OBJECT } jtype;

typedef struct jval {
    jtype t;
    union {
        long long i;
        double d;
        const char *s;
        struct { struct jval **items; const char **keys; size_t n; } c; /* array / object */
    } u;
} jval;

typedef struct json_doc json_doc;

/* Parse; on failure returns NULL and writes a message into err. */
json_doc *json_doc_parse(const char *text, char *err, size_t errlen);
const jval *json_doc_root(const json_doc *doc);
void json_doc_free(json_doc *doc);

static inline int j_is_number(const jval *v) { return v && (v->t == J_INT || v->t == J_REAL); }
static inline int j_is_int(const jval *v) { return v && v->t == J_INT; }
static inline int j_is_string(const jval *v) { return v && v->t == J_STRING; }
static inline int j_is_array(const jval *v) { return v && v->t == J_ARRAY; }
static inline int j_is_object(const jval *v) { return v && v->t == J_OBJECT; }
static inline int j_is_null(const jval *v) { return v && v->t == J_NULL; }
static inline double j_number(const jval *v) { return v->t == J_INT ? (double)v->u.i : v->u.d; }
static inline size_t j_len(const jval *v) { return v->u.c.n; }
static inline const jval *j_at(const jval *v, size_t i) { return v->u.c.items[i]; }
/* object member by key (last duplicate wins), NULL when absent */
const jval *j_get(const jval *obj, const char *key);

/* growing output buffer */
typedef struct { char *p; size_t n, cap; int failed; } jbuf;
void jbuf_init(jbuf *b);
void jbuf_puts(jbuf *b, const char *s);
void jbuf_int(jbuf *b, long long v);
void jbuf_real(jbuf *b, double v);   /* jansson-style real */
char *jbuf_take(jbuf *b);            /* malloc'd NUL-terminated string, or NULL on failure */

#ifdef __cplusplus
}
#endif
#endif
