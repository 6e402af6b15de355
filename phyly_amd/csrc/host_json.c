/* host_json.c -- see host_json.h */
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "host_json.h"

/* ---------------------------------------------------------------- arena */
typedef struct chunk { struct chunk *next; size_t used, cap; } chunk;

struct json_doc {
    chunk *chunks;
    jval *root;
};

typedef struct {
    const char *s, *p;
    json_doc *doc;
    char *err;
    size_t errlen;
    int depth;
    int failed;
} parser;

static void *arena_alloc(json_doc *d, size_t n)
{
    n = (n + 15) & ~(size_t)15;
    chunk *c = d->chunks;
    if (!c || c->used + n > c->cap) {
        size_t cap = n > (1u << 20) ? n : (1u << 20);
        c = malloc(sizeof(chunk) + cap);
        if (!c) return NULL;
        c->next = d->chunks; c->used = 0; c->cap = cap;
        d->chunks = c;
    }
    void *p = (char *)(c + 1) + c->used;
    c->used += n;
    return p;
}

static void fail(parser *ps, const char *msg)
{
    if (ps->failed) return;
    ps->failed = 1;
    if (ps->err && ps->errlen) {
        int line = 1;
        for (const char *q = ps->s; q < ps->p; q++) if (*q == '\n') line++;
        snprintf(ps->err, ps->errlen, "error on json line %d: %s", line, msg);
    }
}

static void skip_ws(parser *ps)
{
    while (*ps->p == ' ' || *ps->p == '\t' || *ps->p == '\n' || *ps->p == '\r') ps->p++;
}

static jval *new_val(parser *ps, jtype t)
{
    jval *v = arena_alloc(ps->doc, sizeof(jval));
    if (!v) { fail(ps, "out of memory"); return NULL; }
    memset(v, 0, sizeof(*v));
    v->t = t;
    return v;
}

static int hex4(const char *p)
{
    int v = 0;
    for (int i = 0; i < 4; i++) {
        char c = p[i];
        v <<= 4;
        if (c >= '0' && c <= '9') v |= c - '0';
        else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
        else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
        else return -1;
    }
    return v;
}

static const char *parse_string_raw(parser *ps)
{
    /* ps->p at opening quote */
    const char *q = ps->p + 1;
    size_t cap = 0;
    for (const char *r = q; *r && *r != '"'; r++) { if (*r == '\\' && r[1]) r++; cap++; }
    char *out = arena_alloc(ps->doc, cap * 3 + 4);
    if (!out) { fail(ps, "out of memory"); return NULL; }
    size_t n = 0;
    while (1) {
        unsigned char c = (unsigned char)*q;
        if (c == 0) { ps->p = q; fail(ps, "premature end of input in string"); return NULL; }
        if (c == '"') { q++; break; }
        if (c < 0x20) { ps->p = q; fail(ps, "control character in string"); return NULL; }
        if (c != '\\') { out[n++] = (char)c; q++; continue; }
        q++;
        switch (*q) {
        case '"': out[n++] = '"'; q++; break;
        case '\\': out[n++] = '\\'; q++; break;
        case '/': out[n++] = '/'; q++; break;
        case 'b': out[n++] = '\b'; q++; break;
        case 'f': out[n++] = '\f'; q++; break;
        case 'n': out[n++] = '\n'; q++; break;
        case 'r': out[n++] = '\r'; q++; break;
        case 't': out[n++] = '\t'; q++; break;
        case 'u': {
            int cp = hex4(q + 1);
            if (cp < 0) { ps->p = q; fail(ps, "invalid unicode escape"); return NULL; }
            q += 5;
            if (cp >= 0xD800 && cp <= 0xDBFF) {
                if (q[0] != '\\' || q[1] != 'u') { ps->p = q; fail(ps, "invalid surrogate pair"); return NULL; }
                int lo = hex4(q + 2);
                if (lo < 0xDC00 || lo > 0xDFFF) { ps->p = q; fail(ps, "invalid surrogate pair"); return NULL; }
                cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                q += 6;
            } else if (cp >= 0xDC00 && cp <= 0xDFFF) { ps->p = q; fail(ps, "invalid unicode escape"); return NULL; }
            if (cp == 0) { ps->p = q; fail(ps, "\\u0000 is not allowed"); return NULL; }
            if (cp < 0x80) out[n++] = (char)cp;
            else if (cp < 0x800) { out[n++] = (char)(0xC0 | (cp >> 6)); out[n++] = (char)(0x80 | (cp & 0x3F)); }
            else if (cp < 0x10000) { out[n++] = (char)(0xE0 | (cp >> 12)); out[n++] = (char)(0x80 | ((cp >> 6) & 0x3F)); out[n++] = (char)(0x80 | (cp & 0x3F)); }
            else { out[n++] = (char)(0xF0 | (cp >> 18)); out[n++] = (char)(0x80 | ((cp >> 12) & 0x3F)); out[n++] = (char)(0x80 | ((cp >> 6) & 0x3F)); out[n++] = (char)(0x80 | (cp & 0x3F)); }
            break;
        }
        default: ps->p = q; fail(ps, "invalid escape"); return NULL;
        }
    }
    out[n] = 0;
    ps->p = q;
    return out;
}

static jval *parse_value(parser *ps);

static jval *parse_number(parser *ps)
{
    const char *q = ps->p;
    int is_real = 0;
    if (*q == '-') q++;
    if (*q == '0') q++;
    else if (*q >= '1' && *q <= '9') { while (*q >= '0' && *q <= '9') q++; }
    else { fail(ps, "invalid token"); return NULL; }
    if (*q == '.') {
        q++; is_real = 1;
        if (!(*q >= '0' && *q <= '9')) { ps->p = q; fail(ps, "invalid number"); return NULL; }
        while (*q >= '0' && *q <= '9') q++;
    }
    if (*q == 'e' || *q == 'E') {
        q++; is_real = 1;
        if (*q == '+' || *q == '-') q++;
        if (!(*q >= '0' && *q <= '9')) { ps->p = q; fail(ps, "invalid number"); return NULL; }
        while (*q >= '0' && *q <= '9') q++;
    }
    jval *v;
    if (!is_real) {
        errno = 0;
        char *end;
        long long i = strtoll(ps->p, &end, 10);
        if (errno == ERANGE || end != q) { fail(ps, "too big integer"); return NULL; }
        v = new_val(ps, J_INT);
        if (v) v->u.i = i;
    } else {
        errno = 0;
        char *end;
        double d = strtod(ps->p, &end);
        if (end != q || (errno == ERANGE && (d == HUGE_VAL || d == -HUGE_VAL))) { fail(ps, "real number overflow"); return NULL; }
        v = new_val(ps, J_REAL);
        if (v) v->u.d = d;
    }
    ps->p = q;
    return v;
}

static jval *parse_container(parser *ps, int is_obj)
{
    jval *v = new_val(ps, is_obj ? J_OBJECT : J_ARRAY);
    if (!v) return NULL;
    if (++ps->depth > 2048) { fail(ps, "maximum parsing depth reached"); return NULL; }
    ps->p++;
    size_t n = 0, cap = 0;
    jval **items = NULL;
    const char **keys = NULL;
    skip_ws(ps);
    const char close = is_obj ? '}' : ']';
    if (*ps->p == close) { ps->p++; }
    else {
        while (1) {
            const char *key = NULL;
            skip_ws(ps);
            if (is_obj) {
                if (*ps->p != '"') { fail(ps, "string or '}' expected"); break; }
                key = parse_string_raw(ps);
                if (!key) break;
                skip_ws(ps);
                if (*ps->p != ':') { fail(ps, "':' expected"); break; }
                ps->p++;
            }
            jval *item = parse_value(ps);
            if (!item) break;
            if (n == cap) {
                size_t ncap = cap ? cap * 2 : 8;
                jval **ni = realloc(items, ncap * sizeof(*ni));
                if (!ni) { fail(ps, "out of memory"); break; }
                items = ni;
                if (is_obj) {
                    const char **nk = realloc((void *)keys, ncap * sizeof(*nk));
                    if (!nk) { fail(ps, "out of memory"); break; }
                    keys = nk;
                }
                cap = ncap;
            }
            items[n] = item;
            if (is_obj) keys[n] = key;
            n++;
            skip_ws(ps);
            if (*ps->p == ',') { ps->p++; continue; }
            if (*ps->p == close) { ps->p++; break; }
            fail(ps, is_obj ? "'}' expected" : "']' expected");
            break;
        }
    }
    ps->depth--;
    if (!ps->failed && n) {
        v->u.c.items = arena_alloc(ps->doc, n * sizeof(jval *));
        if (!v->u.c.items) fail(ps, "out of memory");
        else memcpy(v->u.c.items, items, n * sizeof(jval *));
        if (is_obj && !ps->failed) {
            v->u.c.keys = arena_alloc(ps->doc, n * sizeof(char *));
            if (!v->u.c.keys) fail(ps, "out of memory");
            else memcpy((void *)v->u.c.keys, keys, n * sizeof(char *));
        }
        v->u.c.n = n;
    }
    free(items);
    free((void *)keys);
    return ps->failed ? NULL : v;
}

static jval *parse_value(parser *ps)
{
    skip_ws(ps);
    char c = *ps->p;
    if (c == '{') return parse_container(ps, 1);
    if (c == '[') return parse_container(ps, 0);
    if (c == '"') {
        const char *s = parse_string_raw(ps);
        if (!s) return NULL;
        jval *v = new_val(ps, J_STRING);
        if (v) v->u.s = s;
        return v;
    }
    if (c == '-' || (c >= '0' && c <= '9')) return parse_number(ps);
    if (!strncmp(ps->p, "true", 4)) { ps->p += 4; return new_val(ps, J_TRUE); }
    if (!strncmp(ps->p, "false", 5)) { ps->p += 5; return new_val(ps, J_FALSE); }
    if (!strncmp(ps->p, "null", 4)) { ps->p += 4; return new_val(ps, J_NULL); }
    fail(ps, c ? "invalid token" : "unexpected end of input");
    return NULL;
}

json_doc *json_doc_parse(const char *text, char *err, size_t errlen)
{
    json_doc *doc = calloc(1, sizeof(*doc));
    if (!doc) return NULL;
    parser ps = {text, text, doc, err, errlen, 0, 0};
    skip_ws(&ps);
    if (*ps.p != '{' && *ps.p != '[') fail(&ps, "'[' or '{' expected");
    jval *root = ps.failed ? NULL : parse_value(&ps);
    if (root && !ps.failed) {
        skip_ws(&ps);
        if (*ps.p) fail(&ps, "end of file expected");
    }
    if (ps.failed || !root) { json_doc_free(doc); return NULL; }
    doc->root = root;
    return doc;
}

const jval *json_doc_root(const json_doc *doc) { return doc ? doc->root : NULL; }

void json_doc_free(json_doc *doc)
{
    if (!doc) return;
    chunk *c = doc->chunks;
    while (c) { chunk *n = c->next; free(c); c = n; }
    free(doc);
}

const jval *j_get(const jval *obj, const char *key)
{
    if (!obj || obj->t != J_OBJECT) return NULL;
    for (size_t i = obj->u.c.n; i-- > 0;)
        if (!strcmp(obj->u.c.keys[i], key)) return obj->u.c.items[i];
    return NULL;
}

/* ---------------------------------------------------------------- writer */
void jbuf_init(jbuf *b) { b->p = NULL; b->n = b->cap = 0; b->failed = 0; }

static void jbuf_put(jbuf *b, const char *s, size_t len)
{
    if (b->failed) return;
    if (b->n + len + 1 > b->cap) {
        size_t cap = b->cap ? b->cap * 2 : 256;
        while (cap < b->n + len + 1) cap *= 2;
        char *np = realloc(b->p, cap);
        if (!np) { b->failed = 1; return; }
        b->p = np; b->cap = cap;
    }
    memcpy(b->p + b->n, s, len);
    b->n += len;
    b->p[b->n] = 0;
}

void jbuf_puts(jbuf *b, const char *s) { jbuf_put(b, s, strlen(s)); }

void jbuf_int(jbuf *b, long long v)
{
    char t[32];
    int n = snprintf(t, sizeof t, "%lld", v);
    jbuf_put(b, t, (size_t)n);
}

/* jansson's jsonp_dtostr: "%.17g", force ".0" when integral-looking, strip the
 * '+' and leading zeros of the exponent */
void jbuf_real(jbuf *b, double v)
{
    char t[64];
    int n = snprintf(t, sizeof t, "%.17g", v);
    if (!strchr(t, '.') && !strchr(t, 'e') && !strchr(t, 'n') && !strchr(t, 'i')) { t[n++] = '.'; t[n++] = '0'; t[n] = 0; }
    char *e = strchr(t, 'e');
    if (e) {
        char *src = e + 1, *dst = e + 1;
        if (*src == '-') { src++; dst++; }
        if (*src == '+') src++;
        while (*src == '0' && src[1]) src++;
        memmove(dst, src, strlen(src) + 1);
    }
    jbuf_puts(b, t);
}

char *jbuf_take(jbuf *b)
{
    if (b->failed) { free(b->p); b->p = NULL; return NULL; }
    if (!b->p) { b->p = malloc(1); if (b->p) b->p[0] = 0; }
    char *r = b->p;
    b->p = NULL; b->n = b->cap = 0;
    return r;
}
