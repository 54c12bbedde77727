/* Minimal doubly-linked list used by the est-fact host logic.
 *
 * The reference's algorithms (MEG simplification, embedding enumeration, filters) are defined
 * on linked lists that are modified WHILE being iterated, and their results depend on the exact
 * iterator behaviour of the reference's container (src/list.c, include/list.h:209-280): the
 * iterator caches the NEXT node when it hands out an element, so
 *   - an element appended while the iterator stands on the last node is NOT visited,
 *   - an element appended earlier IS visited,
 *   - "remove at iterator" drops the element handed out last and iteration continues.
 * This container reproduces those semantics (behaviour, not code): circular list with a sentinel,
 * iterator = (next, prev).
 */
#ifndef EF_LIST_H
#define EF_LIST_H

#include <stdbool.h>
#include <stddef.h>
#include <stdlib.h>

typedef struct ef_node { struct ef_node *next, *prev; void* el; } ef_node;
typedef struct ef_list { ef_node sent; size_t size; } ef_list;
typedef struct ef_iter { ef_node *next, *prev; ef_list* l; } ef_iter;

/* ---- 32-byte cells --------------------------------------------------------------------------------
 * An EST goes through a couple of thousand list nodes and list headers; they come from a per-thread
 * free list of 32-byte cells (refilled in blocks, ef_io.c) instead of malloc.  A cell must be
 * returned on the thread that took it, which holds because an EST is processed by one worker from
 * start to end; ef_cell_release_all() gives the thread's blocks back when none of its cells is in
 * use. */
typedef union ef_cell { union ef_cell* next; char bytes[32]; } ef_cell;
extern _Thread_local ef_cell* ef_cell_free_list;
extern _Thread_local long ef_cell_live;
void ef_cell_refill(void);
void ef_cell_release_all(void);
#ifdef EF_NO_CELL_POOL                       /* measurement aid: plain malloc/free */
static inline void* ef_cell_get(void) { return malloc(sizeof(ef_cell)); }
static inline void ef_cell_put(void* p) { free(p); }
#else
static inline void* ef_cell_get(void) {
  if (!ef_cell_free_list) ef_cell_refill();
  ef_cell* c = ef_cell_free_list; ef_cell_free_list = c->next; ++ef_cell_live;
  return c;
}
static inline void ef_cell_put(void* p) {
  ef_cell* c = (ef_cell*)p; c->next = ef_cell_free_list; ef_cell_free_list = c; --ef_cell_live;
}
#endif
typedef char ef_cell_size_check[(sizeof(ef_list) <= sizeof(ef_cell) && sizeof(ef_node) <= sizeof(ef_cell)) ? 1 : -1];
/* the same scheme for 64-byte blocks (MEG vertices) */
typedef union ef_cell64 { union ef_cell64* next; char bytes[64]; } ef_cell64;
extern _Thread_local ef_cell64* ef_cell64_free_list;
void ef_cell64_refill(void);
#ifdef EF_NO_CELL_POOL
static inline void* ef_cell64_get(void) { return malloc(sizeof(ef_cell64)); }
static inline void ef_cell64_put(void* p) { free(p); }
#else
static inline void* ef_cell64_get(void) {
  if (!ef_cell64_free_list) ef_cell64_refill();
  ef_cell64* c = ef_cell64_free_list; ef_cell64_free_list = c->next; ++ef_cell_live;
  return c;
}
static inline void ef_cell64_put(void* p) {
  ef_cell64* c = (ef_cell64*)p; c->next = ef_cell64_free_list; ef_cell64_free_list = c; --ef_cell_live;
}
#endif
#define EFL_NODE_NEW() ((ef_node*)ef_cell_get())
#define EFL_NODE_DEL(n) ef_cell_put(n)

static inline void efl_init(ef_list* l) { l->sent.next = l->sent.prev = &l->sent; l->sent.el = NULL; l->size = 0; }
static inline ef_list* efl_new(void) {
  ef_list* l = (ef_list*)ef_cell_get();
  efl_init(l);
  return l;
}
/* drops the nodes (and, with del, the elements); the header stays usable */
static inline void efl_clear(ef_list* l, void (*del)(void*)) {
  ef_node* n = l->sent.next;
  while (n != &l->sent) { ef_node* nx = n->next; if (del && n->el) del(n->el); EFL_NODE_DEL(n); n = nx; }
  efl_init(l);
}
static inline void efl_free(ef_list* l, void (*del)(void*)) {
  if (!l) return;
  efl_clear(l, del);
  ef_cell_put(l);
}
static inline size_t efl_size(const ef_list* l) { return l->size; }
static inline bool efl_empty(const ef_list* l) { return l->size == 0; }
static inline void efl_push_back(ef_list* l, void* el) {
  ef_node* n = EFL_NODE_NEW();
  n->el = el; n->prev = l->sent.prev; n->next = &l->sent; n->prev->next = n; l->sent.prev = n; ++l->size;
}
static inline void efl_push_front(ef_list* l, void* el) {
  ef_node* n = EFL_NODE_NEW();
  n->el = el; n->next = l->sent.next; n->prev = &l->sent; n->next->prev = n; l->sent.next = n; ++l->size;
}
static inline void* efl_head(const ef_list* l) { return l->sent.next->el; }   /* NULL when empty */
static inline void* efl_tail(const ef_list* l) { return l->sent.prev->el; }
static inline void* efl_pop_front(ef_list* l) {
  ef_node* n = l->sent.next; void* el = n->el;
  l->sent.next = n->next; n->next->prev = &l->sent; EFL_NODE_DEL(n); --l->size; return el;
}
static inline void* efl_pop_back(ef_list* l) {
  ef_node* n = l->sent.prev; void* el = n->el;
  l->sent.prev = n->prev; n->prev->next = &l->sent; EFL_NODE_DEL(n); --l->size; return el;
}

static inline ef_iter efl_begin(ef_list* l) { ef_iter it = { l->sent.next, &l->sent, l }; return it; }
static inline ef_iter efl_end(ef_list* l) { ef_iter it = { &l->sent, l->sent.prev, l }; return it; }
static inline bool efi_has_next(const ef_iter* it) { return it->next != &it->l->sent; }
static inline void* efi_next(ef_iter* it) {
  void* el = it->next->el; it->prev = it->next; it->next = it->next->next; return el;
}
static inline bool efi_has_prev(const ef_iter* it) { return it->prev != &it->l->sent; }
static inline void* efi_prev(ef_iter* it) {
  void* el = it->prev->el; it->next = it->prev; it->prev = it->prev->prev; return el;
}
/* removes the element handed out by the last efi_next */
static inline void efi_remove(ef_iter* it, void (*del)(void*)) {
  ef_node* dead = it->prev;
  it->next->prev = dead->prev; dead->prev->next = it->next;
  if (del) del(dead->el);
  --it->l->size;
  it->prev = dead->prev;
  EFL_NODE_DEL(dead);
}
/* inserts before the element handed out by the last efi_next (include/list.h add_before_iterator) */
static inline void efi_insert_before(ef_iter* it, void* el) {
  ef_node* cur = it->prev;
  ef_node* n = EFL_NODE_NEW();
  n->el = el; n->prev = cur->prev; n->next = cur; cur->prev->next = n; cur->prev = n; ++it->l->size;
}
/* removes the first node holding el; true when found */
static inline bool efl_remove_first(ef_list* l, void* el) {
  for (ef_node* n = l->sent.next; n != &l->sent; n = n->next)
    if (n->el == el) { n->prev->next = n->next; n->next->prev = n->prev; EFL_NODE_DEL(n); --l->size; return true; }
  return false;
}
/* qsort on the element pointers, written back into the existing nodes (as the reference does:
 * same libc qsort, same input order => same order among equal keys) */
static inline void efl_sort(ef_list* l, int (*cmp)(const void*, const void*)) {
  const size_t n = l->size;
  if (n < 2) return;
  void* small[64];
  void** base = n <= 64 ? small : (void**)malloc(n * sizeof(void*));
  size_t i = 0;
  for (ef_node* x = l->sent.next; x != &l->sent; x = x->next) base[i++] = x->el;
  if (n <= 16) {
    /* adjacency lists have a handful of elements: a stable insertion sort (the order glibc's
     * merge-sort qsort gives) without the library call */
    for (size_t a = 1; a < n; ++a) {
      void* key = base[a];
      size_t b = a;
      while (b > 0 && cmp(&base[b - 1], &key) > 0) { base[b] = base[b - 1]; --b; }
      base[b] = key;
    }
  } else {
    qsort(base, n, sizeof(void*), cmp);
  }
  i = 0;
  for (ef_node* x = l->sent.next; x != &l->sent; x = x->next) x->el = base[i++];
  if (base != small) free(base);
}

#endif
